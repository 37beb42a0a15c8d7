// Test driver for integration/ray_tracer_hip.cpp and integration/scene_flatten.hpp, compiled
// against tests/cpp/mock_reference.h.  Reads a flat scene dump written by the Python test,
// rebuilds a pointer-based Scene like scene.cpp would, and either
//   flatten <scene.bin> <out.bin>                      re-flattens it (no GPU needed), or
//   render  <scene.bin> <rays.bin> <out.bin> W H       calls the five ray_tracer_cl.h functions.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mock_reference.h"
#include "scene_flatten.hpp"

static std::vector<float> rdf(FILE *f, size_t n) { std::vector<float> v(n); if (n && fread(v.data(), 4, n, f) != n) abort(); return v; }
static std::vector<int32_t> rdi(FILE *f, size_t n) { std::vector<int32_t> v(n); if (n && fread(v.data(), 4, n, f) != n) abort(); return v; }
static glm::vec3 v3(const std::vector<float> &a, size_t i) { return glm::vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

static void loadScene(Scene &s, const char *path)
{
    FILE *f = fopen(path, "rb"); if (!f) abort();
    auto hdr = rdi(f, 5);
    size_t nt = hdr[0], ns = hdr[1], nm = hdr[2], nx = hdr[3], nl = hdr[4];
    auto tp = rdf(f, 3 * nt), te1 = rdf(f, 3 * nt), te2 = rdf(f, 3 * nt), tn = rdf(f, 3 * nt), tuv = rdf(f, 6 * nt);
    auto tm = rdi(f, nt);
    auto sc = rdf(f, 3 * ns), sr = rdf(f, ns); auto sm = rdi(f, ns);
    auto mc = rdf(f, 3 * nm), kd = rdf(f, nm), ks = rdf(f, nm), hd = rdf(f, nm), kr = rdf(f, nm); auto mt = rdi(f, nm);
    std::vector<std::shared_ptr<tex_image>> tex;
    for (size_t i = 0; i < nx; i++) {
        auto wh = rdi(f, 2);
        auto t = std::make_shared<tex_image>(); t->w = wh[0]; t->h = wh[1]; t->data.resize((size_t)wh[0] * wh[1] * 4);
        if (fread(t->data.data(), 1, t->data.size(), f) != t->data.size()) abort();
        tex.push_back(t);
    }
    auto lp = rdf(f, 3 * nl), li = rdf(f, nl), ld = rdf(f, nl);
    auto cam = rdf(f, 3), cm = rdf(f, 9), hz = rdf(f, 3), am = rdf(f, 3);
    fclose(f);
    for (size_t i = 0; i < nm; i++) {
        auto m = std::make_shared<Material>();
        m->color = v3(mc, i); m->diffusiveFactor = kd[i]; m->specularFactor = ks[i]; m->specularHardness = hd[i]; m->reflectionFactor = kr[i];
        if (mt[i] >= 0) { m->texImage = tex[mt[i]]; m->textured = true; }
        s.materials.push_back(m);
    }
    auto mat = [&](int idx) { return idx >= 0 ? s.materials[idx] : std::make_shared<Material>(); };   // scene.cpp:87-89
    for (size_t i = 0; i < nt; i++) {
        // the construction of scene.cpp:90-93
        std::shared_ptr<Triangle> t(new Triangle(v3(tp, i), v3(te1, i), v3(te2, i), glm::vec2(tuv[6 * i], tuv[6 * i + 1]),
                                                 glm::vec2(tuv[6 * i + 2], tuv[6 * i + 3]), glm::vec2(tuv[6 * i + 4], tuv[6 * i + 5]),
                                                 v3(tn, i), mat(tm[i])));
        s.triangles.push_back(t);
    }
    for (size_t i = 0; i < ns; i++) s.spheres.push_back(std::shared_ptr<Sphere>(new Sphere(v3(sc, i), sr[i], mat(sm[i]))));      // scene.cpp:107
    for (size_t i = 0; i < nl; i++) s.lamps.push_back(std::shared_ptr<Lamp>(new Lamp(v3(lp, i), li[i], ld[i])));                  // scene.cpp:117
    s.camPos = v3(cam, 0);
    for (int i = 0; i < 3; i++) s.camMat[i] = glm::vec3(cm[3 * i], cm[3 * i + 1], cm[3 * i + 2]);
    s.worldHorizonColor = v3(hz, 0); s.worldAmbientColor = v3(am, 0);
}

template <class T> static void wr(FILE *f, const T *p, size_t n) { if (n) fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    Scene scene;
    loadScene(scene, argv[2]);
    if (!strcmp(argv[1], "flatten")) {
        rt_hip::FlatScene<Scene> fs(scene);
        const hipexec_scene_desc &d = fs.desc();
        FILE *f = fopen(argv[3], "wb");
        int32_t hdr[5] = {(int32_t)d.n_tri, (int32_t)d.n_sph, (int32_t)d.n_mat, (int32_t)d.n_tex, (int32_t)d.n_lamp};
        wr(f, hdr, 5);
        wr(f, d.tri_p, 3 * d.n_tri); wr(f, d.tri_e1, 3 * d.n_tri); wr(f, d.tri_e2, 3 * d.n_tri); wr(f, d.tri_norm, 3 * d.n_tri);
        wr(f, d.tri_uv, 6 * d.n_tri); wr(f, d.tri_mat, d.n_tri);
        wr(f, d.sph_center, 3 * d.n_sph); wr(f, d.sph_radius, d.n_sph); wr(f, d.sph_mat, d.n_sph);
        wr(f, d.mat_color, 3 * d.n_mat); wr(f, d.mat_kd, d.n_mat); wr(f, d.mat_ks, d.n_mat); wr(f, d.mat_hard, d.n_mat);
        wr(f, d.mat_kr, d.n_mat); wr(f, d.mat_tex, d.n_mat);
        for (uint32_t i = 0; i < d.n_tex; i++) {
            int32_t wh[2] = {(int32_t)d.tex_w[i], (int32_t)d.tex_h[i]};
            wr(f, wh, 2); wr(f, d.tex_rgba + d.tex_offset[i], (size_t)wh[0] * wh[1] * 4);
        }
        wr(f, d.lamp_pos, 3 * d.n_lamp); wr(f, d.lamp_intensity, d.n_lamp); wr(f, d.lamp_distance, d.n_lamp);
        wr(f, d.cam_pos, 3); wr(f, d.cam_mat, 9); wr(f, d.horizon, 3); wr(f, d.ambient, 3);
        fclose(f);
        return 0;
    }
    if (!strcmp(argv[1], "render") && argc >= 7) {
        int W = atoi(argv[5]), H = atoi(argv[6]);
        FILE *rf = fopen(argv[3], "rb"); if (!rf) abort();
        auto nr = rdi(rf, 1); auto rays6 = rdf(rf, 6 * (size_t)nr[0]); fclose(rf);
        std::vector<RayData> rays;
        for (int i = 0; i < nr[0]; i++) rays.push_back(RayData(rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2], rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]));
        image_bitmap full(W, H), blocks(W, H);
        renderSceneCl(full, scene);
        // the block walk of ray_tracer_cl.cpp:15-55 with SUB_BLOCK_WIDTH x SUB_BLOCK_HEIGHT (48x48) blocks, through renderSubBlockCl
        extern std::shared_ptr<OpenClExecutor> makeExecutorForTest(const Scene &);
        auto ex = makeExecutorForTest(scene);
        for (int y = 0; y < H; y += SUB_BLOCK_HEIGHT) for (int x = 0; x < W; x += SUB_BLOCK_WIDTH)
            renderSubBlockCl(blocks, scene, ex, x, y, std::min(SUB_BLOCK_WIDTH, W - x), std::min(SUB_BLOCK_HEIGHT, H - y), W, H);
        std::vector<Hit> hits;
        computeClosestHitsCl(scene, ex, rays, hits);
        std::vector<char> occ(rays.size(), 0);
        for (size_t i = 0; i < occ.size(); i += 5) occ[i] = 1;               // pre-set entries must survive the OR
        computeAnyHitsCl(scene, ex, rays, occ.data());
        std::vector<glm::vec3> cols(rays.size(), glm::vec3(0.5f, 0.25f, 0.125f));
        traceRaysCl(scene, ex, rays, cols, 0);
        FILE *f = fopen(argv[4], "wb");
        wr(f, static_cast<float *>(full.getRawData()), (size_t)W * H * 3);
        wr(f, static_cast<float *>(blocks.getRawData()), (size_t)W * H * 3);
        for (size_t i = 0; i < rays.size(); i++) {
            float rec[12] = {hits[i].isHit ? 1.f : 0.f, hits[i].t, hits[i].point.x, hits[i].point.y, hits[i].point.z,
                             hits[i].norm.x, hits[i].norm.y, hits[i].norm.z, hits[i].color.x, hits[i].color.y, hits[i].color.z,
                             hits[i].isHit ? hits[i].mtl->diffusiveFactor : -1.f};
            wr(f, rec, 12);
        }
        wr(f, occ.data(), occ.size());
        for (size_t i = 0; i < rays.size(); i++) { float c[3] = {cols[i].x, cols[i].y, cols[i].z}; wr(f, c, 3); }
        // the same executor handed a scene with another camera (ray_tracer_cl.cpp:80-92 makes the rays from the scene of the call)
        Scene moved = scene;
        moved.camPos = glm::vec3(scene.camPos.x + 0.75f, scene.camPos.y + 0.5f, scene.camPos.z - 1.25f);
        image_bitmap blocks2(W, H);
        for (int y = 0; y < H; y += SUB_BLOCK_HEIGHT) for (int x = 0; x < W; x += SUB_BLOCK_WIDTH)
            renderSubBlockCl(blocks2, moved, ex, x, y, std::min(SUB_BLOCK_WIDTH, W - x), std::min(SUB_BLOCK_HEIGHT, H - y), W, H);
        wr(f, static_cast<float *>(blocks2.getRawData()), (size_t)W * H * 3);
        fclose(f);
        return 0;
    }
    return 2;
}
