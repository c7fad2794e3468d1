// gltf.cpp — minimal glTF 2.0 reader for exactly what the reference's loader consumes
// (scene::load_from_gltf and read_mesh, src/scene.rs:116-250): .gltf JSON with external or base64 buffers and
// .glb containers; per node the LOCAL transform (the reference ignores the hierarchy, src/scene.rs:213-216),
// the first TRIANGLES primitive of the node's mesh (src/scene.rs:127-131) with POSITION (transformed),
// NORMAL (not transformed, src/scene.rs:177-179), TEXCOORD_0 and indices; the material reduced to "emissive
// -> white DiffuseLight of the emissive strength, else Lambertian of the base colour" (src/scene.rs:136-159);
// perspective cameras with focus distance 10 and no defocus (src/scene.rs:223-234).
// Host only; exposed through the rt_gltf_* entry points.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "jpeg_decode.h"
#include "png_decode.h"

namespace rt {
void set_last_error(const std::string &message); // api.cpp: the message rt_last_error() returns
}

namespace {

// ---------------------------------------------------------------------------------------------------------
// JSON
// ---------------------------------------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *find(const char *key) const {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char *key) const { return find(key) != nullptr; }
    const Json &at(const char *key) const {
        const Json *j = find(key);
        if (!j) throw std::runtime_error(std::string("missing key '") + key + "'");
        return *j;
    }
    const Json &at(size_t i) const {
        if (kind != Arr || i >= arr.size()) throw std::runtime_error("array index out of range");
        return arr[i];
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
    double number() const {
        if (kind != Num) throw std::runtime_error("expected a number");
        return num;
    }
    size_t index() const {
        double v = number();
        if (v < 0 || v != std::floor(v)) throw std::runtime_error("expected a non-negative integer");
        return (size_t)v;
    }
};

struct JsonParser {
    const char *p, *end;
    explicit JsonParser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
    [[noreturn]] void fail(const char *msg) { throw std::runtime_error(std::string("JSON: ") + msg); }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    Json parse_value(int depth = 0) {
        if (depth > 200) fail("nesting too deep");
        ws();
        if (p >= end) fail("unexpected end");
        Json j;
        switch (*p) {
        case '{': {
            j.kind = Json::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return j; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') fail("expected a key");
                std::string key = parse_string();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                j.obj.emplace_back(std::move(key), parse_value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return j; }
                fail("expected ',' or '}'");
            }
        }
        case '[': {
            j.kind = Json::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(parse_value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return j; }
                fail("expected ',' or ']'");
            }
        }
        case '"':
            j.kind = Json::Str;
            j.str = parse_string();
            return j;
        case 't':
            if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; j.kind = Json::Bool; j.b = true; return j; }
            fail("bad literal");
        case 'f':
            if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; j.kind = Json::Bool; return j; }
            fail("bad literal");
        case 'n':
            if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return j; }
            fail("bad literal");
        default: {
            const char *q = p;
            while (q < end && (isdigit((unsigned char)*q) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E')) ++q;
            if (q == p) fail("unexpected character");
            std::string tok(p, q);
            char *e = nullptr;
            j.num = strtod(tok.c_str(), &e);
            if (!e || *e) fail("bad number");
            j.kind = Json::Num;
            p = q;
            return j;
        }
        }
    }
    std::string parse_string() {
        std::string out;
        ++p; // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': { // keep ASCII, replace the rest: names are not used for anything but display
                    if (end - p < 5) fail("bad \\u escape");
                    unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    out += v < 128 ? (char)v : '?';
                    p += 4;
                    break;
                }
                default: out += *p; break;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return out;
    }
};

std::string read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::ios_base::failure("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string base64_decode(const char *s, size_t n) {
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        memset(table, -1, sizeof table);
        const char *al = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) table[(unsigned char)al[i]] = (int8_t)i;
        init = true;
    }
    std::string out;
    out.reserve(n * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        int8_t v = table[(unsigned char)s[i]];
        if (v < 0) continue; // padding / whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out += (char)((acc >> bits) & 0xFF);
        }
    }
    return out;
}

std::string dirname_of(const std::string &path) {
    size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

// ---------------------------------------------------------------------------------------------------------
// document access
// ---------------------------------------------------------------------------------------------------------
struct Document {
    Json root;
    std::vector<std::string> buffers;
    std::string dir;

    void load(const std::string &path) {
        dir = dirname_of(path);
        std::string raw = read_file(path);
        std::string bin_chunk;
        std::string json_text;
        if (raw.size() >= 12 && !memcmp(raw.data(), "glTF", 4)) { // GLB container
            uint32_t length;
            memcpy(&length, raw.data() + 8, 4);
            size_t off = 12;
            while (off + 8 <= raw.size() && off < length) {
                uint32_t clen, ctype;
                memcpy(&clen, raw.data() + off, 4);
                memcpy(&ctype, raw.data() + off + 4, 4);
                if (off + 8 + clen > raw.size()) throw std::runtime_error("truncated GLB chunk");
                if (ctype == 0x4E4F534Au) json_text.assign(raw.data() + off + 8, clen);
                else if (ctype == 0x004E4942u) bin_chunk.assign(raw.data() + off + 8, clen);
                off += 8 + clen;
            }
        } else {
            json_text.swap(raw);
        }
        JsonParser parser(json_text);
        root = parser.parse_value();
        if (root.kind != Json::Obj) throw std::runtime_error("glTF root is not an object");
        if (const Json *bufs = root.find("buffers")) {
            for (const Json &b : bufs->arr) {
                if (const Json *uri = b.find("uri")) {
                    const std::string &u = uri->str;
                    if (u.rfind("data:", 0) == 0) {
                        size_t comma = u.find(',');
                        if (comma == std::string::npos) throw std::runtime_error("bad data URI");
                        buffers.push_back(base64_decode(u.data() + comma + 1, u.size() - comma - 1));
                    } else {
                        buffers.push_back(read_file(dir + "/" + u));
                    }
                } else {
                    buffers.push_back(bin_chunk);
                }
            }
        }
    }

    // The encoded file of image `index`: a data URI, a file next to the document, or a buffer view.
    std::string image_bytes(size_t index) const {
        const Json &img = root.at("images").at(index);
        if (const Json *uri = img.find("uri")) {
            const std::string &u = uri->str;
            if (u.rfind("data:", 0) == 0) {
                size_t comma = u.find(',');
                if (comma == std::string::npos) throw std::runtime_error("bad data URI");
                return base64_decode(u.data() + comma + 1, u.size() - comma - 1);
            }
            return read_file(dir + "/" + u);
        }
        const Json &bv = root.at("bufferViews").at(img.at("bufferView").index());
        const std::string &buf = buffers.at(bv.at("buffer").index());
        const size_t start = bv.has("byteOffset") ? bv.at("byteOffset").index() : 0;
        const size_t len = bv.at("byteLength").index();
        if (start + len > buf.size()) throw std::runtime_error("image buffer view exceeds its buffer");
        return buf.substr(start, len);
    }

    // Reads accessor `index` as rows of `want_comps` values converted by `conv(component_type, ptr)`.
    template <class T, class Conv> std::vector<T> read_accessor(size_t index, size_t want_comps, Conv conv) const {
        const Json &a = root.at("accessors").at(index);
        if (a.has("sparse")) throw std::runtime_error("sparse accessors are not supported");
        const std::string &type = a.at("type").str;
        size_t comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
        if (comps != want_comps) throw std::runtime_error("accessor has unexpected type " + type);
        const unsigned ctype = (unsigned)a.at("componentType").index();
        const size_t csize = (ctype == 5120 || ctype == 5121) ? 1 : (ctype == 5122 || ctype == 5123) ? 2 : 4;
        const size_t count = a.at("count").index();
        std::vector<T> out(count * comps);
        if (!a.has("bufferView")) return out; // all zeros
        const Json &bv = root.at("bufferViews").at(a.at("bufferView").index());
        const std::string &buf = buffers.at(bv.at("buffer").index());
        size_t start = (bv.has("byteOffset") ? bv.at("byteOffset").index() : 0) +
                       (a.has("byteOffset") ? a.at("byteOffset").index() : 0);
        size_t elem = csize * comps;
        size_t stride = bv.has("byteStride") ? bv.at("byteStride").index() : elem;
        if (stride == 0) stride = elem;
        if (count && start + (count - 1) * stride + elem > buf.size()) throw std::runtime_error("accessor exceeds its buffer");
        for (size_t i = 0; i < count; ++i)
            for (size_t c = 0; c < comps; ++c)
                out[i * comps + c] = conv(ctype, buf.data() + start + i * stride + c * csize);
        return out;
    }
};

float to_f32(unsigned ctype, const char *p, bool normalized_ints) {
    switch (ctype) {
    case 5126: { float f; memcpy(&f, p, 4); return f; }
    case 5121: { uint8_t v; memcpy(&v, p, 1); return normalized_ints ? (float)v / 255.0f : (float)v; }
    case 5123: { uint16_t v; memcpy(&v, p, 2); return normalized_ints ? (float)v / 65535.0f : (float)v; }
    default: throw std::runtime_error("unsupported float component type");
    }
}

uint32_t to_u32(unsigned ctype, const char *p) {
    switch (ctype) {
    case 5121: { uint8_t v; memcpy(&v, p, 1); return v; }
    case 5123: { uint16_t v; memcpy(&v, p, 2); return v; }
    case 5125: { uint32_t v; memcpy(&v, p, 4); return v; }
    default: throw std::runtime_error("unsupported index component type");
    }
}

// Column-major 4x4 of the node's local transform: "matrix" as given, else T * R * S composed in binary32 like
// the gltf crate's Transform::matrix (rotation terms x2 = x + x, xx2 = x2 * x, ...).
void node_matrix(const Json &node, float m[16]) {
    if (const Json *mat = node.find("matrix")) {
        for (int i = 0; i < 16; ++i) m[i] = (float)mat->at(i).number();
        return;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (const Json *j = node.find("translation")) for (int i = 0; i < 3; ++i) t[i] = (float)j->at(i).number();
    if (const Json *j = node.find("rotation")) for (int i = 0; i < 4; ++i) q[i] = (float)j->at(i).number();
    if (const Json *j = node.find("scale")) for (int i = 0; i < 3; ++i) s[i] = (float)j->at(i).number();
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const float sx2 = x2 * w, sy2 = y2 * w, sz2 = z2 * w;
    const float r[3][3] = {{1.0f - yy2 - zz2, xy2 - sz2, xz2 + sy2},
                           {xy2 + sz2, 1.0f - xx2 - zz2, yz2 - sx2},
                           {xz2 - sy2, yz2 + sx2, 1.0f - xx2 - yy2}};
    memset(m, 0, 16 * sizeof(float));
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) m[c * 4 + rr] = r[rr][c] * s[c];
    m[12] = t[0];
    m[13] = t[1];
    m[14] = t[2];
    m[15] = 1.0f;
}

struct MeshData {
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices;
    int32_t material = 0;
};

} // namespace

struct rt_gltf {
    std::vector<MeshData> mesh_data;
    std::vector<rt_mesh> meshes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<std::vector<uint8_t>> texels; // decoded images the IMAGE textures point into
    std::vector<rt_camera_settings> cameras;
    rt_scene_desc desc{};
};

namespace {

int add_solid(rt_gltf &g, float r, float gr, float b) {
    rt_texture t{};
    t.kind = RT_TEX_SOLID;
    t.rgb[0] = r;
    t.rgb[1] = gr;
    t.rgb[2] = b;
    t.even = t.odd = -1;
    g.textures.push_back(t);
    return (int)g.textures.size() - 1;
}

// `load_image` (src/scene.rs:87-114) on the image the gltf crate decoded (PNG or JPEG, told apart by their signatures
// as the crate's `guess_format` does): 8-bit RGB texels, (l, l, l) for grey images.
struct DecodedImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

DecodedImage decode_image(const uint8_t *bytes, size_t size) {
    DecodedImage img;
    std::string err;
    const bool jpeg = size >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8;
    const bool ok = jpeg ? rt::decode_jpeg_rgb8(bytes, size, img.width, img.height, img.rgb, err)
                         : rt::decode_png_rgb8(bytes, size, img.width, img.height, img.rgb, err);
    if (!ok) throw std::runtime_error("base-colour texture: " + err);
    return img;
}

DecodedImage decode_image(const std::string &file) {
    return decode_image(reinterpret_cast<const uint8_t *>(file.data()), file.size());
}

int add_image(rt_gltf &g, DecodedImage &&img) {
    g.texels.push_back(std::move(img.rgb));
    rt_texture t{};
    t.kind = RT_TEX_IMAGE;
    t.even = t.odd = -1;
    t.width = img.width;
    t.height = img.height;
    t.pixels = g.texels.back().data();
    g.textures.push_back(t);
    return (int)g.textures.size() - 1;
}

void load_into(const std::string &path, rt_gltf &g) {
    Document doc;
    doc.load(path);
    const Json &root = doc.root;
    const Json *nodes = root.find("nodes");
    if (!nodes) return;
    for (const Json &node : nodes->arr) {
        float m[16];
        node_matrix(node, m);
        // glam Affine3A::from_mat4: upper 3x4 as x_axis, y_axis, z_axis, translation
        const float aff[12] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10], m[12], m[13], m[14]};
        if (const Json *mi = node.find("mesh")) {
            const Json &mesh = root.at("meshes").at(mi->index());
            const Json *prim = nullptr;
            if (const Json *prims = mesh.find("primitives"))
                for (const Json &p : prims->arr) {
                    size_t mode = p.has("mode") ? p.at("mode").index() : 4;
                    if (mode == 4) { prim = &p; break; }
                }
            if (!prim) throw std::runtime_error("mesh must have at least one triangles primitive");
            static const Json empty_obj = [] { Json j; j.kind = Json::Obj; return j; }();
            const Json &mat = prim->has("material") ? root.at("materials").at(prim->at("material").index()) : empty_obj;
            const Json &pbr = mat.has("pbrMetallicRoughness") ? mat.at("pbrMetallicRoughness") : empty_obj;
            float emissive[3] = {0, 0, 0};
            if (const Json *e = mat.find("emissiveFactor")) for (int i = 0; i < 3; ++i) emissive[i] = (float)e->at(i).number();
            rt_material rm{};
            rm.left = rm.right = -1;
            // The reference decodes the base-colour image before it looks at the emissive factor
            // (src/scene.rs:138-149), so an undecodable image fails the load even on an emitter.
            DecodedImage image;
            const Json *bct = pbr.find("baseColorTexture");
            if (bct) {
                const Json &tex = root.at("textures").at(bct->at("index").index());
                image = decode_image(doc.image_bytes(tex.at("source").index()));
            }
            if (emissive[0] != 0.0f || emissive[1] != 0.0f || emissive[2] != 0.0f) {
                float strength = 1.0f;
                if (const Json *ext = mat.find("extensions"))
                    if (const Json *es = ext->find("KHR_materials_emissive_strength"))
                        if (const Json *v = es->find("emissiveStrength")) strength = (float)v->number();
                rm.kind = RT_MAT_DIFFUSE_LIGHT;
                rm.texture = add_solid(g, 1.0f, 1.0f, 1.0f);
                rm.param = strength;
            } else if (bct) {
                rm.kind = RT_MAT_LAMBERTIAN;
                rm.texture = add_image(g, std::move(image));
            } else {
                float col[3] = {1.0f, 1.0f, 1.0f};
                if (const Json *c = pbr.find("baseColorFactor")) for (int i = 0; i < 3; ++i) col[i] = (float)c->at(i).number();
                rm.kind = RT_MAT_LAMBERTIAN;
                rm.texture = add_solid(g, col[0], col[1], col[2]);
            }
            g.materials.push_back(rm);
            MeshData md;
            md.material = (int32_t)g.materials.size() - 1;
            const Json &attrs = prim->has("attributes") ? prim->at("attributes") : empty_obj;
            if (const Json *a = attrs.find("POSITION")) {
                std::vector<float> p = doc.read_accessor<float>(a->index(), 3, [](unsigned ct, const char *q) { return to_f32(ct, q, false); });
                md.positions.resize(p.size());
                for (size_t i = 0; i + 2 < p.size(); i += 3) { // glam Affine3A::transform_point3a
                    for (int k = 0; k < 3; ++k) {
                        float r = aff[k] * p[i];
                        r = r + aff[3 + k] * p[i + 1];
                        r = r + aff[6 + k] * p[i + 2];
                        md.positions[i + k] = r + aff[9 + k];
                    }
                }
            }
            if (const Json *ix = prim->find("indices")) {
                md.indices = doc.read_accessor<uint32_t>(ix->index(), 1, to_u32);
                md.indices.resize(md.indices.size() / 3 * 3); // chunks(3) with a short tail would panic in the reference
            }
            if (const Json *a = attrs.find("NORMAL"))
                md.normals = doc.read_accessor<float>(a->index(), 3, [](unsigned ct, const char *q) { return to_f32(ct, q, false); });
            if (const Json *a = attrs.find("TEXCOORD_0"))
                md.uvs = doc.read_accessor<float>(a->index(), 2, [](unsigned ct, const char *q) { return to_f32(ct, q, true); });
            g.mesh_data.push_back(std::move(md));
        }
        if (const Json *ci = node.find("camera")) {
            const Json &cam = root.at("cameras").at(ci->index());
            if (cam.has("type") && cam.at("type").str == "perspective") {
                const Json &p = cam.at("perspective");
                rt_camera_settings cs{};
                cs.y_fov = (float)p.at("yfov").number();
                cs.z_near = (float)p.at("znear").number();
                cs.z_far = p.has("zfar") ? (float)p.at("zfar").number() : std::numeric_limits<float>::infinity();
                memcpy(cs.transform, aff, sizeof aff);
                cs.focus_dist = 10.0f;
                cs.defocus_angle = 0.0f;
                g.cameras.push_back(cs);
            }
        }
    }
    g.meshes.resize(g.mesh_data.size());
    for (size_t i = 0; i < g.mesh_data.size(); ++i) {
        const MeshData &md = g.mesh_data[i];
        rt_mesh &rm = g.meshes[i];
        rm.num_vertices = (uint32_t)(md.positions.size() / 3);
        rm.positions = md.positions.data();
        rm.num_normals = (uint32_t)(md.normals.size() / 3);
        rm.normals = md.normals.empty() ? nullptr : md.normals.data();
        rm.num_uvs = (uint32_t)(md.uvs.size() / 2);
        rm.uvs = md.uvs.empty() ? nullptr : md.uvs.data();
        rm.num_faces = (uint32_t)(md.indices.size() / 3);
        rm.indices = md.indices.empty() ? nullptr : md.indices.data();
        rm.material = md.material;
    }
    g.desc.num_meshes = (uint32_t)g.meshes.size();
    g.desc.meshes = g.meshes.data();
    g.desc.num_materials = (uint32_t)g.materials.size();
    g.desc.materials = g.materials.data();
    g.desc.num_textures = (uint32_t)g.textures.size();
    g.desc.textures = g.textures.data();
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) int rt_gltf_load(const char *path, rt_gltf **out) {
    if (!path || !out) {
        rt::set_last_error("rt_gltf_load: null argument");
        return RT_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    std::unique_ptr<rt_gltf> g(new rt_gltf());
    try {
        load_into(path, *g);
    } catch (const std::ios_base::failure &e) {
        rt::set_last_error(e.what());
        return RT_ERR_IO;
    } catch (const std::bad_alloc &) {
        rt::set_last_error("out of memory");
        return RT_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        rt::set_last_error(e.what());
        return RT_ERR_PARSE;
    }
    *out = g.release();
    return RT_OK;
}

__attribute__((visibility("default"))) const rt_scene_desc *rt_gltf_scene(const rt_gltf *g) { return g ? &g->desc : nullptr; }

__attribute__((visibility("default"))) uint32_t rt_gltf_camera_count(const rt_gltf *g) { return g ? (uint32_t)g->cameras.size() : 0u; }

__attribute__((visibility("default"))) int rt_gltf_camera(const rt_gltf *g, uint32_t index, rt_camera_settings *out) {
    if (!g || !out) {
        rt::set_last_error("rt_gltf_camera: null argument");
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (index < g->cameras.size()) {
        *out = g->cameras[index];
        return RT_OK;
    }
    return rt_camera_settings_default(out); // SceneDescription::camera falls back to the default (src/scene.rs:25-27)
}

__attribute__((visibility("default"))) int rt_gltf_free(rt_gltf *g) {
    delete g;
    return RT_OK;
}

__attribute__((visibility("default"))) int rt_image_decode(const void *bytes, size_t size, uint32_t *width, uint32_t *height,
                                                           uint8_t **pixels) {
    if (!bytes || !width || !height || !pixels) {
        rt::set_last_error("rt_image_decode: null argument");
        return RT_ERR_INVALID_ARGUMENT;
    }
    *pixels = nullptr;
    try {
        DecodedImage img = decode_image(static_cast<const uint8_t *>(bytes), size);
        uint8_t *out = static_cast<uint8_t *>(malloc(img.rgb.empty() ? 1 : img.rgb.size()));
        if (!out) throw std::bad_alloc();
        memcpy(out, img.rgb.data(), img.rgb.size());
        *width = img.width;
        *height = img.height;
        *pixels = out;
    } catch (const std::bad_alloc &) {
        rt::set_last_error("out of memory");
        return RT_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        rt::set_last_error(e.what());
        return RT_ERR_PARSE;
    }
    return RT_OK;
}

__attribute__((visibility("default"))) void rt_image_free(uint8_t *pixels) { free(pixels); }

} // extern "C"
