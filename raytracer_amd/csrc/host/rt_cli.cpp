// rt_cli — command-line driver with the reference CLI's flags (src/main.rs:38-65):
//   rt_cli [--bvh-disabled] [--debug] [-W|--width N] [-H|--height N] [-d|--max-depth N] [--spp N]
//          [-c|--camera N] [--seed N] [--batch] [--device N] [--gpus N] INPUT.gltf [OUTPUT.(png|ppm|pfm)]
// Like `main` it loads the glTF file, builds the BVH, picks the camera and renders progressively
// (src/main.rs:91-104); instead of streaming to a tev viewer it writes the last report to OUTPUT.  `--batch`
// runs `Renderer::render` (linear mean) instead.  `--bvh-disabled` is accepted and ignored, as in the reference.
// `--gpus N` splits every pass's samples over devices 0..N-1 of the node (one all-reduce per pass).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rt_host.hpp"

int main(int argc, char **argv) {
    using namespace rt::host;
    RenderSettings rs;
    bool debug = false, batch = false;
    int device = 0, gpus = 0;
    std::string input, output = "image.ppm";
    int positional = 0;
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) {
            std::fprintf(stderr, "missing value for %s\n", argv[i]);
            std::exit(2);
        }
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--bvh-disabled") {
        } else if (a == "--debug") debug = true;
        else if (a == "--batch") batch = true;
        else if (a == "-W" || a == "--width") rs.image_width = (uint32_t)std::strtoul(need(i), nullptr, 10);
        else if (a == "-H" || a == "--height") rs.image_height = (uint32_t)std::strtoul(need(i), nullptr, 10);
        else if (a == "-d" || a == "--max-depth") rs.max_depth = (uint32_t)std::strtoul(need(i), nullptr, 10);
        else if (a == "--spp") rs.samples_per_pixel = (uint32_t)std::strtoul(need(i), nullptr, 10);
        else if (a == "-c" || a == "--camera") rs.selected_camera = std::strtoul(need(i), nullptr, 10);
        else if (a == "--seed") rs.seed = std::strtoull(need(i), nullptr, 10);
        else if (a == "--device") device = std::atoi(need(i));
        else if (a == "--gpus") gpus = std::atoi(need(i));
        else if (a == "-h" || a == "--help") {
            std::printf("usage: rt_cli [--bvh-disabled] [--debug] [-W N] [-H N] [-d N] [--spp N] [-c N] [--seed N] [--batch] "
                        "[--device N] [--gpus N] INPUT.gltf [OUTPUT.png|OUTPUT.ppm|OUTPUT.pfm]\n");
            return 0;
        } else if (!a.empty() && a[0] == '-') {
            std::fprintf(stderr, "unknown flag %s\n", a.c_str());
            return 2;
        } else if (positional++ == 0) input = a;
        else output = a;
    }
    if (input.empty()) {
        std::fprintf(stderr, "error: the INPUT argument is required\n");
        return 2;
    }
    try {
        using clock = std::chrono::steady_clock;
        auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        auto t0 = clock::now();
        SceneDescription desc = SceneDescription::load_from_gltf(input);
        auto t1 = clock::now();
        if (gpus > 0) { // one process, several GPUs
            rt_camera camera = make_camera(desc.camera(rs.selected_camera), rs.image_width, rs.image_height);
            NodeRenderer renderer(gpus, desc.desc(), camera, rs);
            auto t2 = clock::now();
            std::printf("loaded %s in %.1f ms; building the BVH on %d GPUs took %.1f ms\n", input.c_str(), ms(t0, t1), gpus, ms(t1, t2));
            ImageOutput out{output};
            if (batch) {
                out.write(renderer.render(16), rs.image_width, rs.image_height);
            } else {
                std::vector<float> last;
                renderer.render_progressive([&](const std::vector<float> &px, uint32_t, uint32_t, uint32_t sample) {
                    if (debug) std::printf("Outputted image after sample %u\n", sample);
                    last = px;
                });
                out.write(last, rs.image_width, rs.image_height);
            }
            std::printf("Rendering took %.1f ms on %d GPUs; wrote %s\n", ms(t2, clock::now()), gpus, output.c_str());
            return 0;
        }
        Context ctx(device);
        Scene scene(ctx, desc.desc());
        auto t2 = clock::now();
        rt_scene_info info = scene.info();
        std::printf("loaded %s in %.1f ms; building BVH took %.1f ms (%u triangles, %u nodes, depth %u)\n", input.c_str(),
                    ms(t0, t1), ms(t1, t2), info.num_triangles, info.num_nodes, info.depth);
        std::printf("extents of the scene: [%g %g %g] .. [%g %g %g]\n", info.bounds_min[0], info.bounds_min[1], info.bounds_min[2],
                    info.bounds_max[0], info.bounds_max[1], info.bounds_max[2]);
        rt_camera camera = make_camera(desc.camera(rs.selected_camera), rs.image_width, rs.image_height);
        Renderer renderer(ctx, camera, scene, rs);
        ImageOutput out{output};
        rt_stats stats{};
        auto t3 = clock::now();
        if (batch) {
            std::vector<float> img = renderer.render(16, &stats);
            out.write(img, rs.image_width, rs.image_height);
        } else {
            std::vector<float> last;
            renderer.render_progressive(
                [&](const std::vector<float> &px, uint32_t, uint32_t, uint32_t sample) {
                    if (debug) std::printf("Outputted image after sample %u\n", sample);
                    last = px;
                },
                16, &stats);
            out.write(last, rs.image_width, rs.image_height);
        }
        auto t4 = clock::now();
        std::printf("Rendering took %.1f ms (%llu paths, %llu rays, %.1f Mrays/s on the GPU); wrote %s\n", ms(t3, t4),
                    (unsigned long long)stats.paths, (unsigned long long)stats.rays_traced,
                    stats.total_ms > 0 ? stats.rays_traced / stats.total_ms / 1e3 : 0.0, output.c_str());
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
