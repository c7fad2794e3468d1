// rt_host.hpp — C++ host-side mirror of the reference's `Renderer` surface over the C ABI.
//
// The reference is compiled code (Rust) whose toolchain is not available to this build, so the host layer a
// Rust caller would get from a `-sys` crate is provided in C++ with the reference's names and argument
// meaning: `load_from_gltf` (src/scene.rs:208), `SceneDescription::camera` (src/scene.rs:25-27), `Camera::new`
// (src/camera.rs:20), `RenderSettings` (src/scene.rs:53-60), `Renderer::{new, render, render_progressive}`
// (src/renderer.rs:25,72,105) and `ImageOutput` (src/renderer.rs:178-226, file sink only).
// Errors surface as rt::host::Error carrying the rt_status code and rt_last_error().
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../../include/rt_mi355x.h"

namespace rt {
namespace host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

inline void check(int rc) {
    if (rc != RT_OK) throw Error(rc, rt_last_error());
}

// `RenderSettings` (src/scene.rs:53-60); defaults are the CLI's (src/main.rs:46-59,86).
struct RenderSettings {
    uint32_t image_width = 1280;
    uint32_t image_height = 720;
    size_t selected_camera = 0;
    uint32_t max_depth = 50;
    uint32_t samples_per_pixel = 100;
    float background_color[3] = {0.0f, 0.0f, 0.0f};
    uint64_t seed = 1; // the reference seeds from OS entropy; a fixed seed makes runs reproducible
};

class Context {
  public:
    explicit Context(int device = 0, uint32_t flags = 0) { check(rt_context_create(device, nullptr, flags, &h_)); }
    ~Context() { rt_context_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rt_context *get() const { return h_; }

  private:
    rt_context *h_ = nullptr;
};

// `SceneDescription` (src/scene.rs:18-22) as loaded from a glTF file; owns the host copy.
class SceneDescription {
  public:
    static SceneDescription load_from_gltf(const std::string &path) {
        SceneDescription s;
        check(rt_gltf_load(path.c_str(), &s.g_));
        return s;
    }
    SceneDescription(SceneDescription &&o) noexcept : g_(o.g_) { o.g_ = nullptr; }
    SceneDescription &operator=(SceneDescription &&o) noexcept {
        if (this != &o) {
            rt_gltf_free(g_);
            g_ = o.g_;
            o.g_ = nullptr;
        }
        return *this;
    }
    ~SceneDescription() { rt_gltf_free(g_); }
    const rt_scene_desc *desc() const { return rt_gltf_scene(g_); }
    size_t camera_count() const { return rt_gltf_camera_count(g_); }
    // `SceneDescription::camera(index)`: beyond the file's cameras the default camera.
    rt_camera_settings camera(size_t index) const {
        rt_camera_settings cs;
        check(rt_gltf_camera(g_, (uint32_t)index, &cs));
        return cs;
    }

  private:
    SceneDescription() = default;
    rt_gltf *g_ = nullptr;
};

// The scene after `build_bvh(BvhType::Tree)` (src/scene.rs:29-49), resident on the GPU.
class Scene {
  public:
    Scene(const Context &ctx, const rt_scene_desc *desc) { check(rt_scene_create(ctx.get(), desc, &h_)); }
    ~Scene() { rt_scene_destroy(h_); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    rt_scene *get() const { return h_; }
    rt_scene_info info() const {
        rt_scene_info i;
        check(rt_scene_get_info(h_, &i));
        return i;
    }

  private:
    rt_scene *h_ = nullptr;
};

// `Camera::new(settings, width, height)` (src/camera.rs:20-64).
inline rt_camera make_camera(const rt_camera_settings &settings, uint32_t width, uint32_t height) {
    rt_camera cam;
    check(rt_camera_from_settings(&settings, width, height, &cam));
    return cam;
}

// src/renderer.rs:167-171
inline bool should_save_image(uint32_t current_sample, uint32_t sample_count) {
    return current_sample <= 5 || current_sample % 16 == 0 || current_sample == sample_count;
}

// `pixels_to_image` (src/renderer.rs:173-176): Rgb32F -> Rgb8, round(clamp(c, 0, 1) * 255), NaN -> 0.
inline std::vector<uint8_t> pixels_to_rgb8(const std::vector<float> &pixels) {
    std::vector<uint8_t> out(pixels.size());
    for (size_t i = 0; i < pixels.size(); ++i) {
        float c = pixels[i];
        if (!(c == c)) c = 0.0f;
        c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        out[i] = (uint8_t)std::lround(c * 255.0f);
    }
    return out;
}

// 8-bit RGB PNG (colour type 2, filter 0 on every row, one IDAT chunk deflated by zlib).
inline std::vector<uint8_t> encode_png_rgb8(const std::vector<uint8_t> &rgb, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw((size_t)height * ((size_t)width * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *row = raw.data() + (size_t)y * ((size_t)width * 3 + 1);
        row[0] = 0;
        std::memcpy(row + 1, rgb.data() + (size_t)y * width * 3, (size_t)width * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw Error(RT_ERR_IO, "zlib compress failed");
    z.resize(zlen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto be32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s));
    };
    auto chunk = [&](const char *tag, const std::vector<uint8_t> &body) {
        be32((uint32_t)body.size());
        const size_t start = out.size();
        out.insert(out.end(), tag, tag + 4);
        out.insert(out.end(), body.begin(), body.end());
        be32((uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> ihdr;
    for (uint32_t v : {width, height})
        for (int s = 24; s >= 0; s -= 8) ihdr.push_back((uint8_t)(v >> s));
    for (uint8_t b : {8, 2, 0, 0, 0}) ihdr.push_back(b);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    return out;
}

// File sink of `ImageOutput` (src/renderer.rs:178-226): the reference converts to 8 bits with pixels_to_image
// and lets the image crate pick the format from the extension.  Here `.png` and `.ppm` (any other extension)
// hold those 8-bit values and `.pfm` keeps the floats.
struct ImageOutput {
    std::string path;
    void write(const std::vector<float> &pixels, uint32_t width, uint32_t height) const {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw Error(RT_ERR_IO, "cannot open " + path);
        const bool pfm = path.size() >= 4 && path.compare(path.size() - 4, 4, ".pfm") == 0;
        const bool png = path.size() >= 4 && path.compare(path.size() - 4, 4, ".png") == 0;
        if (png) {
            std::vector<uint8_t> file = encode_png_rgb8(pixels_to_rgb8(pixels), width, height);
            std::fwrite(file.data(), 1, file.size(), f);
        } else if (pfm) {
            std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
            for (uint32_t j = height; j-- > 0;) // PFM rows run bottom to top
                std::fwrite(pixels.data() + (size_t)j * width * 3, sizeof(float), (size_t)width * 3, f);
        } else {
            std::vector<uint8_t> px = pixels_to_rgb8(pixels);
            std::fprintf(f, "P6\n%u %u\n255\n", width, height);
            std::fwrite(px.data(), 1, px.size(), f);
        }
        std::fclose(f);
    }
};

// `Renderer` (src/renderer.rs:18-157).
class Renderer {
  public:
    // `Renderer::new(camera, scene, render)`
    Renderer(const Context &ctx, rt_camera camera, const Scene &scene, RenderSettings render)
        : ctx_(ctx), camera_(camera), scene_(scene), render_(render) {}

    // `Renderer::render(square_size)`: linear mean of all samples.  Returns row-major RGB floats; the reference
    // converts them with pixels_to_image.  `square_size` shaped the reference's CPU chunks and is ignored.
    std::vector<float> render(size_t /*square_size*/ = 16, rt_stats *stats = nullptr) const {
        std::vector<float> sums((size_t)render_.image_width * render_.image_height * 3);
        rt_settings st = settings(RT_MODE_LINEAR_MEAN, 0, render_.samples_per_pixel);
        check(rt_render(ctx_.get(), scene_.get(), &camera_, &st, sums.data(), stats));
        const float sample_scale = 1.0f / (float)render_.samples_per_pixel;
        for (float &c : sums) c = c * sample_scale;
        return sums;
    }

    // `Renderer::render_progressive(output, square_size)`: running sum of gamma-mapped samples; `output` gets
    // aggregate * (1 / current_sample) after samples 1-5, every 16th and the last.
    void render_progressive(const std::function<void(const std::vector<float> &, uint32_t, uint32_t, uint32_t)> &output,
                            size_t /*square_size*/ = 16, rt_stats *total = nullptr) const {
        const uint32_t w = render_.image_width, h = render_.image_height, spp = render_.samples_per_pixel;
        // The running aggregate_image (src/renderer.rs:110) lives on the GPU: rt_render_device adds the samples of a
        // report interval to it one by one in ascending order, the order of src/renderer.rs:135-137, so the sums do
        // not depend on where the intervals fall; each report downloads a copy.
        std::vector<float> aggregate((size_t)w * h * 3, 0.0f), image(aggregate.size());
        float *d_aggregate = nullptr;
        check(rt_accum_create(ctx_.get(), w, h, &d_aggregate));
        struct Release {
            rt_context *ctx;
            float *p;
            ~Release() { (void)rt_accum_destroy(ctx, p); }
        } release{ctx_.get(), d_aggregate};
        uint32_t done = 0;
        while (done < spp) {
            uint32_t next = done + 1;
            while (!should_save_image(next, spp)) ++next;
            rt_settings st = settings(RT_MODE_GAMMA_PER_SAMPLE, done, next - done);
            rt_stats s;
            check(rt_render_device(ctx_.get(), scene_.get(), &camera_, &st, d_aggregate, &s));
            check(rt_accum_download(ctx_.get(), d_aggregate, aggregate.size(), aggregate.data()));
            if (total) {
                total->rays_traced += s.rays_traced;
                total->paths += s.paths;
                total->total_ms += s.total_ms;
            }
            done = next;
            const float sample_scale = 1.0f / (float)done;
            for (size_t i = 0; i < aggregate.size(); ++i) image[i] = aggregate[i] * sample_scale;
            output(image, w, h, done);
        }
    }

  private:
    rt_settings settings(uint32_t mode, uint32_t begin, uint32_t count) const {
        rt_settings st{};
        st.image_width = render_.image_width;
        st.image_height = render_.image_height;
        st.max_depth = render_.max_depth;
        st.samples_per_pixel = render_.samples_per_pixel;
        for (int i = 0; i < 3; ++i) st.background_color[i] = render_.background_color[i];
        st.mode = mode;
        st.seed = render_.seed;
        st.sample_begin = begin;
        st.sample_count = count;
        return st;
    }
    const Context &ctx_;
    rt_camera camera_;
    const Scene &scene_;
    RenderSettings render_;
};

// The same renderer over several GPUs of one node, driven by this one process: the scene is replicated on
// devices 0 .. n_gpus-1, every pass splits its samples over the GPUs (`rt_render_node`) and one RCCL
// all-reduce adds the accumulators.  Not in the reference, which is single-device.
class NodeRenderer {
  public:
    NodeRenderer(int n_gpus, const rt_scene_desc *desc, rt_camera camera, RenderSettings render, uint32_t flags = 0)
        : camera_(camera), render_(render) {
        for (int d = 0; d < n_gpus; ++d) {
            ctxs_.push_back(std::make_unique<Context>(d, flags));
            scenes_.push_back(std::make_unique<Scene>(*ctxs_.back(), desc));
            raw_ctxs_.push_back(ctxs_.back()->get());
            raw_scenes_.push_back(scenes_.back()->get());
        }
    }

    std::vector<float> render(size_t /*square_size*/ = 16) const {
        std::vector<float> sums = pass(RT_MODE_LINEAR_MEAN, 0, render_.samples_per_pixel);
        const float sample_scale = 1.0f / (float)render_.samples_per_pixel;
        for (float &c : sums) c = c * sample_scale;
        return sums;
    }

    void render_progressive(const std::function<void(const std::vector<float> &, uint32_t, uint32_t, uint32_t)> &output,
                            size_t /*square_size*/ = 16) const {
        const uint32_t w = render_.image_width, h = render_.image_height, spp = render_.samples_per_pixel;
        std::vector<float> aggregate((size_t)w * h * 3, 0.0f);
        uint32_t done = 0;
        while (done < spp) {
            uint32_t next = done + 1;
            while (!should_save_image(next, spp)) ++next;
            std::vector<float> part = pass(RT_MODE_GAMMA_PER_SAMPLE, done, next - done);
            for (size_t i = 0; i < aggregate.size(); ++i) aggregate[i] += part[i];
            done = next;
            const float sample_scale = 1.0f / (float)done;
            std::vector<float> image(aggregate.size());
            for (size_t i = 0; i < aggregate.size(); ++i) image[i] = aggregate[i] * sample_scale;
            output(image, w, h, done);
        }
    }

  private:
    std::vector<float> pass(uint32_t mode, uint32_t begin, uint32_t count) const {
        rt_settings st{};
        st.image_width = render_.image_width;
        st.image_height = render_.image_height;
        st.max_depth = render_.max_depth;
        st.samples_per_pixel = render_.samples_per_pixel;
        for (int i = 0; i < 3; ++i) st.background_color[i] = render_.background_color[i];
        st.mode = mode;
        st.seed = render_.seed;
        st.sample_begin = begin;
        st.sample_count = count;
        std::vector<float> sums((size_t)st.image_width * st.image_height * 3);
        check(rt_render_node(raw_ctxs_.data(), raw_scenes_.data(), (int)raw_ctxs_.size(), &camera_, &st, sums.data()));
        return sums;
    }
    std::vector<std::unique_ptr<Context>> ctxs_;
    std::vector<std::unique_ptr<Scene>> scenes_;
    std::vector<rt_context *> raw_ctxs_;
    std::vector<const rt_scene *> raw_scenes_;
    rt_camera camera_;
    RenderSettings render_;
};

} // namespace host
} // namespace rt
