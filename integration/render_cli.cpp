// render_cli.cpp — headless renderer: loads a `.scene` file and renders it with the HIP executor.
// Stands in for main.cpp:22-117 (which needs GLFW/OpenGL) where there is no display:
//
//   hiprt_render scene.scene -o out.png [--width 800 --height 600] [--mode cl|cpu] [--depth D]
//                [--algo bvh|scan] [--frames N] [--linear] [--ids ids.bin] [--dump-scene scene.bin]
//
// Defaults follow config.h:8-16 (800x600, the OpenCL path's rules, no reflection).  Output by
// extension: .png / .ppm (8 bit, floor(clamp(c,0,1)*255+0.5), top row first) or .pfm (float, bottom
// row first = the order of image_bitmap, image_bitmap.h:22,45).  --dump-scene only loads the scene,
// writes its flattened arrays and exits (needs no GPU).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hip_executor.h"
#include "scene_loader.hpp"

namespace {

int usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s scene.scene [-o out.png|out.ppm|out.pfm] [--width W] [--height H] [--mode cl|cpu]\n"
            "          [--depth D] [--algo bvh|scan] [--frames N] [--optimize|--no-optimize] [--linear] [--ids ids.bin]\n"
            "          [--dump-scene file]\n"
            "  --optimize: rebuild the hierarchy once with the surface-area topology (hipexec_optimize: same picture, faster\n"
            "              frames, a few ms per 100 000 triangles); the default from 16 frames up\n",
            argv0);
    return 2;
}

bool ends_with(const std::string &s, const char *suffix) {
    size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace

int main(int argc, char **argv)
{
    std::string scene_path, out_path, ids_path, dump_path;
    int width = 800, height = 600, mode = HIPEXEC_MODE_CL, depth = 0, frames = 1, flags = 0;
    bool linear = false;
    int optimize = -1;              // -1: by the number of frames
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-o") out_path = next("-o");
        else if (a == "--width") width = atoi(next("--width"));
        else if (a == "--height") height = atoi(next("--height"));
        else if (a == "--depth") depth = atoi(next("--depth"));
        else if (a == "--frames") frames = atoi(next("--frames"));
        else if (a == "--optimize") optimize = 1;
        else if (a == "--no-optimize") optimize = 0;
        else if (a == "--ids") ids_path = next("--ids");
        else if (a == "--dump-scene") dump_path = next("--dump-scene");
        else if (a == "--linear") linear = true;
        else if (a == "--mode") {
            std::string m = next("--mode");
            if (m == "cl") mode = HIPEXEC_MODE_CL; else if (m == "cpu") mode = HIPEXEC_MODE_CPU; else return usage(argv[0]);
        } else if (a == "--algo") {
            std::string m = next("--algo");
            if (m == "bvh") flags |= HIPEXEC_FLAG_BVH; else if (m == "scan") flags |= HIPEXEC_FLAG_SCAN; else return usage(argv[0]);
        } else if (a == "-h" || a == "--help") return usage(argv[0]);
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); return usage(argv[0]); }
        else scene_path = a;
    }
    if (scene_path.empty()) return usage(argv[0]);
    try {
        auto t0 = std::chrono::steady_clock::now();
        rt_hip::LoadedScene scene = rt_hip::LoadedScene::from_file(scene_path);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "loaded %s: %u triangles, %zu spheres, %zu lamps, %zu materials, %zu textures (%.1f ms)\n",
                scene_path.c_str(), scene.n_tri(), scene.sph_radius.size(), scene.lamp_intensity.size(),
                scene.mat_kd.size(), scene.tex_w.size(), std::chrono::duration<double, std::milli>(t1 - t0).count());
        if (!dump_path.empty()) { scene.dump(dump_path); return 0; }

        hipexec_scene_desc desc = scene.desc();
        hipexec_t *ex = nullptr;
        if (hipexec_create(&ex, &desc, -1) != HIPEXEC_OK) { fprintf(stderr, "hipexec_create: %s\n", hipexec_last_error()); return 1; }
        if (optimize == 1 || (optimize < 0 && frames >= 16)) {
            float ms = 0.0f;
            if (hipexec_optimize(ex, &ms) != HIPEXEC_OK) { fprintf(stderr, "hipexec_optimize: %s\n", hipexec_last_error()); hipexec_destroy(ex); return 1; }
            fprintf(stderr, "hierarchy rebuilt with the surface-area topology (%.1f ms)\n", ms);
        }
        hipexec_render_cfg cfg{width, height, mode, depth, linear ? 0 : 1, 0, 1, flags | HIPEXEC_FLAG_TIME_KERNELS};
        const int spp = mode == HIPEXEC_MODE_CL ? 1 : 4;
        std::vector<float> rgb((size_t)width * height * 3);
        std::vector<int32_t> ids(ids_path.empty() ? 0 : (size_t)width * height * spp);
        hipexec_stats st{};
        for (int f = 0; f < frames; f++) {
            auto a = std::chrono::steady_clock::now();
            if (hipexec_render(ex, &cfg, rgb.data(), ids.empty() ? nullptr : ids.data(), &st) != HIPEXEC_OK) {
                fprintf(stderr, "hipexec_render: %s\n", hipexec_last_error());
                hipexec_destroy(ex);
                return 1;
            }
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
            unsigned long long rays = st.rays_primary + st.rays_bounce + st.rays_shadow;
            // the reference prints the frame time in ms (main.cpp:102-104)
            printf("frame %d: %.3f ms (device %.3f ms), %llu rays, %.1f Mrays/s\n", f, ms, st.total_ms, rays, rays / (st.total_ms * 1e3));
        }
        hipexec_destroy(ex);

        if (!ids_path.empty()) {
            FILE *f = fopen(ids_path.c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", ids_path.c_str()); return 1; }
            fwrite(ids.data(), sizeof(int32_t), ids.size(), f);
            fclose(f);
        }
        if (!out_path.empty()) {
            if (ends_with(out_path, ".pfm")) {
                FILE *f = fopen(out_path.c_str(), "wb");
                if (!f) { fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
                fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
                fwrite(rgb.data(), sizeof(float), rgb.size(), f);
                fclose(f);
            } else {
                std::vector<uint8_t> px((size_t)width * height * 3);
                for (int y = 0; y < height; y++)
                    for (int x = 0; x < width * 3; x++) {
                        float c = rgb[(size_t)(height - 1 - y) * width * 3 + x];      // row 0 of the frame is the bottom row
                        c = c > 0.0f ? (c < 1.0f ? c : 1.0f) : 0.0f;
                        px[(size_t)y * width * 3 + x] = (uint8_t)floorf(c * 255.0f + 0.5f);
                    }
                if (ends_with(out_path, ".ppm")) {
                    FILE *f = fopen(out_path.c_str(), "wb");
                    if (!f) { fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
                    fprintf(f, "P6\n%d %d\n255\n", width, height);
                    fwrite(px.data(), 1, px.size(), f);
                    fclose(f);
                } else {
                    rt_hip::write_png_rgb8(out_path, (uint32_t)width, (uint32_t)height, px.data());
                }
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
