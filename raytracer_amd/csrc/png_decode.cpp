// png_decode.cpp — PNG -> 8-bit RGB texels for base-colour textures.
//
// The reference gets its texels from the gltf crate (which decodes PNG files with the image crate) and maps
// the result through `load_image` (src/scene.rs:87-114): 8-bit grey, grey+alpha, RGB and RGBA are accepted,
// everything else (16-bit samples) panics.  `Image::value_at` (src/texture.rs:115-136) then reads channels
// 0..2 of `DynamicImage::get_pixel`, i.e. (l, l, l) for grey images and (r, g, b) otherwise; alpha is never
// used.  PNG is lossless, so the texels below are the reference's texels: palette images are expanded to RGB
// and 1/2/4-bit grey is scaled to 0..255, as the decoder's EXPAND transformation does.
// zlib does the inflate; chunk CRCs are not verified.
#include "png_decode.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>

namespace rt {

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the per-scanline filter of one (sub)image in place; `rows` points at its first filter byte.
bool unfilter(uint8_t *rows, uint32_t height, size_t row_bytes, size_t bpp, std::string &err) {
    const size_t stride = row_bytes + 1;
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *cur = rows + (size_t)y * stride + 1;
        const uint8_t *up = y ? cur - stride : nullptr;
        const uint8_t type = cur[-1];
        for (size_t i = 0; i < row_bytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred;
            switch (type) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: pred = paeth(a, b, c); break;
            default: err = "PNG: unknown filter type"; return false;
            }
            cur[i] = (uint8_t)(cur[i] + pred);
        }
    }
    return true;
}

} // namespace

bool decode_png_rgb8(const uint8_t *data, size_t n, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb,
                     std::string &err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(data, sig, 8) != 0) {
        err = "not a PNG file";
        return false;
    }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> palette, idat;
    bool have_ihdr = false, done = false;
    for (size_t off = 8; off + 12 <= n && !done;) {
        const uint32_t len = be32(data + off);
        const uint8_t *type = data + off + 4, *body = data + off + 8;
        if (len > n - off - 12) {
            err = "PNG: truncated chunk";
            return false;
        }
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) {
                err = "PNG: short IHDR";
                return false;
            }
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) {
                err = "PNG: unknown compression, filter or interlace method";
                return false;
            }
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        off += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) {
        err = "PNG: missing header or unreasonable size";
        return false;
    }
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "PNG: unknown colour type"; return false;
    }
    if (depth == 16) { // load_image's `_ => panic!("unsupported image format ...")` arm
        err = "unsupported image format: 16-bit samples (the reference accepts 8-bit R, RG, RGB, RGBA only)";
        return false;
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && depth == 8);
    if (!depth_ok) {
        err = "PNG: bit depth not allowed for the colour type";
        return false;
    }
    if (ctype == 3 && palette.size() < 3) {
        err = "PNG: palette image without PLTE";
        return false;
    }
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = (bits_pp + 7) / 8; // filter distance in bytes

    // the seven Adam7 passes, or the one pass of a non-interlaced image
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass *passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < n_passes; ++p) {
        const uint32_t pw = (w > passes[p].x0) ? (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const uint32_t ph = (h > passes[p].y0) ? (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (pw && ph) raw_size += (size_t)ph * (1 + (pw * bits_pp + 7) / 8);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    const int zrc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || got != raw_size) {
        err = "PNG: image data does not inflate to the size the header implies";
        return false;
    }
    rgb.assign((size_t)w * h * 3, 0);
    const int grey_scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    size_t pos = 0;
    for (int p = 0; p < n_passes; ++p) {
        const uint32_t pw = (w > passes[p].x0) ? (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const uint32_t ph = (h > passes[p].y0) ? (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (!pw || !ph) continue;
        const size_t row_bytes = (pw * bits_pp + 7) / 8;
        if (!unfilter(raw.data() + pos, ph, row_bytes, bpp, err)) return false;
        for (uint32_t y = 0; y < ph; ++y) {
            const uint8_t *row = raw.data() + pos + (size_t)y * (row_bytes + 1) + 1;
            for (uint32_t x = 0; x < pw; ++x) {
                uint8_t s[4] = {0, 0, 0, 0};
                if (depth == 8) {
                    for (int c = 0; c < channels; ++c) s[c] = row[(size_t)x * channels + c];
                } else { // 1, 2 or 4 bits, one channel, most significant bits first
                    const size_t bit = (size_t)x * depth;
                    s[0] = (uint8_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                }
                uint8_t *out = rgb.data() + ((size_t)(passes[p].y0 + y * passes[p].dy) * w + passes[p].x0 + (size_t)x * passes[p].dx) * 3;
                if (ctype == 3) {
                    if ((size_t)s[0] * 3 + 2 >= palette.size()) {
                        err = "PNG: palette index out of range";
                        return false;
                    }
                    memcpy(out, palette.data() + (size_t)s[0] * 3, 3);
                } else if (ctype == 0 || ctype == 4) {
                    out[0] = out[1] = out[2] = (uint8_t)(ctype == 0 ? s[0] * grey_scale : s[0]);
                } else {
                    out[0] = s[0];
                    out[1] = s[1];
                    out[2] = s[2];
                }
            }
        }
        pos += (size_t)ph * (row_bytes + 1);
    }
    width = w;
    height = h;
    return true;
}

} // namespace rt
