// scene_loader.hpp — reads a `.scene` JSON file (the format loadScene parses, scene.cpp:15-135,
// written by scene_exporter.py:11-137) straight into the flattened arrays of
// hipexec_scene_desc, without the reference's Scene classes, glm or GLFW.
//
// Reproduced behaviour of loadScene:
//   * "world": null -> ambient (0,0,0), horizon (0.3,0.3,0.3)                 (scene.cpp:24-35)
//   * materials: scaleX/scaleY are read only for materials with an imagePath   (scene.cpp:44-49)
//   * faces: p = v0, e1 = v1 - v0, e2 = v2 - v0, norm = normalize(cross(e1, e2)); UVs relative to
//     the first corner and multiplied by (texW * scaleX, texH * scaleY)         (scene.cpp:61-94)
//   * sphere centres and lamp positions are parsed into int (truncation)       (scene.cpp:97-99,112-114)
//   * missing / null material -> index -1 (default Material(), scene.h:32-44)
// All arithmetic is binary32 in the reference's order; compile host code without FMA contraction.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "hip_executor.h"
#include "json_min.hpp"
#include "png_min.hpp"

namespace rt_hip {

class LoadedScene {
public:
    std::vector<float> tri_p, tri_e1, tri_e2, tri_norm, tri_uv;
    std::vector<int32_t> tri_mat;
    std::vector<float> sph_center, sph_radius;
    std::vector<int32_t> sph_mat;
    std::vector<float> mat_color, mat_kd, mat_ks, mat_hard, mat_kr;
    std::vector<int32_t> mat_tex;
    std::vector<uint8_t> tex_rgba;
    std::vector<uint64_t> tex_offset;
    std::vector<uint32_t> tex_w, tex_h;
    std::vector<float> lamp_pos, lamp_intensity, lamp_distance;
    float cam_pos[3] = {0, 0, 0}, cam_mat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float horizon[3] = {0.3f, 0.3f, 0.3f}, ambient[3] = {0, 0, 0};
    float ambient_factor = 0.0f;

    static LoadedScene from_file(const std::string &path) {
        std::vector<uint8_t> bytes = read_file(path);
        std::string dir = ".";
        size_t slash = path.find_last_of('/');
        if (slash != std::string::npos) dir = path.substr(0, slash);
        return from_json(Json::parse(std::string(bytes.begin(), bytes.end())), dir);
    }

    static LoadedScene from_json(const Json &doc, const std::string &base_dir) {
        LoadedScene s;
        const Json &world = doc.get("world");
        if (!world.is_null()) {
            vec3(world.need("ambientColor"), s.ambient);
            vec3(world.need("horizonColor"), s.horizon);
            if (!world.get("ambientFactor").is_null()) s.ambient_factor = world.get("ambientFactor").f32();     // scene.cpp:30
        }
        const Json &mats = doc.get("materials");
        std::vector<float> scale_x, scale_y;
        for (size_t i = 0; i < mats.size(); i++) {
            const Json &m = mats.at(i);
            float c[3]; vec3(m.need("diffusiveColor"), c);
            s.mat_color.insert(s.mat_color.end(), c, c + 3);
            s.mat_kd.push_back(m.need("diffusiveFactor").f32());
            s.mat_ks.push_back(m.need("specularFactor").f32());
            s.mat_hard.push_back(m.need("specularHardness").f32());
            s.mat_kr.push_back(m.need("reflectionFactor").f32());
            const Json &ip = m.get("imagePath");
            if (!ip.is_null()) {
                std::string p = ip.string();
                if (p.empty() || p[0] != '/') p = base_dir + "/" + p;
                ImageRGBA8 img = decode_png_file(p);
                s.mat_tex.push_back((int32_t)s.tex_w.size());
                s.tex_offset.push_back(s.tex_rgba.size());
                s.tex_w.push_back(img.width); s.tex_h.push_back(img.height);
                s.tex_rgba.insert(s.tex_rgba.end(), img.rgba.begin(), img.rgba.end());
                scale_x.push_back(m.need("scaleX").f32());
                scale_y.push_back(m.need("scaleY").f32());
            } else {
                s.mat_tex.push_back(-1); scale_x.push_back(1.0f); scale_y.push_back(1.0f);
            }
        }
        const Json &verts = doc.get("vertices");
        std::vector<float> v(3 * verts.size());
        for (size_t i = 0; i < verts.size(); i++) vec3(verts.at(i), &v[3 * i]);
        const Json &faces = doc.get("faces");
        const size_t n = faces.size();
        s.tri_p.resize(3 * n); s.tri_e1.resize(3 * n); s.tri_e2.resize(3 * n); s.tri_norm.resize(3 * n);
        s.tri_uv.assign(6 * n, 0.0f); s.tri_mat.assign(n, -1);
        for (size_t i = 0; i < n; i++) {
            const Json &f = faces.at(i);
            const Json &iv = f.need("vertices");
            const float *a[3];
            for (int k = 0; k < 3; k++) {
                double idx = iv.at(k).number();
                if (idx < 0 || idx >= (double)verts.size()) throw std::runtime_error("scene: face vertex index out of range");
                a[k] = &v[3 * (size_t)idx];
            }
            float *p = &s.tri_p[3 * i], *e1 = &s.tri_e1[3 * i], *e2 = &s.tri_e2[3 * i], *nm = &s.tri_norm[3 * i];
            for (int k = 0; k < 3; k++) { p[k] = a[0][k]; e1[k] = a[1][k] - a[0][k]; e2[k] = a[2][k] - a[0][k]; }
            const Json &mi = f.get("material");
            int mat = mi.is_null() ? -1 : (int)mi.number();
            if (mat >= (int)mats.size()) throw std::runtime_error("scene: face material index out of range");
            s.tri_mat[i] = mat < 0 ? -1 : mat;
            const Json &uv = f.get("uv");
            if (!uv.is_null()) {
                float c[3][2];
                for (int k = 0; k < 3; k++) { c[k][0] = uv.at(k).at(0).f32(); c[k][1] = uv.at(k).at(1).f32(); }
                float *t = &s.tri_uv[6 * i];
                t[0] = c[0][0]; t[1] = c[0][1];
                t[2] = c[1][0] - c[0][0]; t[3] = c[1][1] - c[0][1];
                t[4] = c[2][0] - c[0][0]; t[5] = c[2][1] - c[0][1];
            }
            if (mat >= 0 && s.mat_tex[mat] >= 0) {
                int tx = s.mat_tex[mat];
                float sx = (float)s.tex_w[tx] * scale_x[mat], sy = (float)s.tex_h[tx] * scale_y[mat];
                float *t = &s.tri_uv[6 * i];
                t[0] *= sx; t[2] *= sx; t[4] *= sx;
                t[1] *= sy; t[3] *= sy; t[5] *= sy;
            }
            // normalize(cross(e1, e2)) with glm's operation order
            float cx = e1[1] * e2[2] - e2[1] * e1[2], cy = e1[2] * e2[0] - e2[2] * e1[0], cz = e1[0] * e2[1] - e2[0] * e1[1];
            float inv = 1.0f / sqrtf((cx * cx + cy * cy) + cz * cz);
            nm[0] = cx * inv; nm[1] = cy * inv; nm[2] = cz * inv;
        }
        const Json &sph = doc.get("spheres");
        for (size_t i = 0; i < sph.size(); i++) {
            const Json &sp = sph.at(i);
            for (int k = 0; k < 3; k++) s.sph_center.push_back((float)sp.need("center").at(k).i32());
            s.sph_radius.push_back(sp.need("radius").f32());
            const Json &mi = sp.get("material");
            int mat = mi.is_null() ? -1 : (int)mi.number();
            if (mat >= (int)mats.size()) throw std::runtime_error("scene: sphere material index out of range");
            s.sph_mat.push_back(mat < 0 ? -1 : mat);
        }
        const Json &lamps = doc.get("lamps");
        for (size_t i = 0; i < lamps.size(); i++) {
            const Json &lp = lamps.at(i);
            for (int k = 0; k < 3; k++) s.lamp_pos.push_back((float)lp.need("pos").at(k).i32());
            s.lamp_intensity.push_back(lp.need("intensity").f32());
            s.lamp_distance.push_back(lp.need("distance").f32());
        }
        vec3(doc.need("translate"), s.cam_pos);
        const Json &tr = doc.need("transform");
        for (int r = 0; r < 3; r++) vec3(tr.at(r), &s.cam_mat[3 * r]);
        return s;
    }

    uint32_t n_tri() const { return (uint32_t)tri_mat.size(); }

    // descriptor over this object's storage (valid while the object lives and is not modified)
    hipexec_scene_desc desc() const {
        hipexec_scene_desc d{};
        d.n_tri = n_tri();
        d.tri_p = tri_p.data(); d.tri_e1 = tri_e1.data(); d.tri_e2 = tri_e2.data(); d.tri_norm = tri_norm.data();
        d.tri_uv = tri_uv.data(); d.tri_mat = tri_mat.data();
        d.n_sph = (uint32_t)sph_radius.size();
        d.sph_center = sph_center.data(); d.sph_radius = sph_radius.data(); d.sph_mat = sph_mat.data();
        d.n_mat = (uint32_t)mat_kd.size();
        d.mat_color = mat_color.data(); d.mat_kd = mat_kd.data(); d.mat_ks = mat_ks.data(); d.mat_hard = mat_hard.data();
        d.mat_kr = mat_kr.data(); d.mat_tex = mat_tex.data();
        d.n_tex = (uint32_t)tex_w.size();
        d.tex_rgba = tex_rgba.data(); d.tex_offset = tex_offset.data(); d.tex_w = tex_w.data(); d.tex_h = tex_h.data();
        d.n_lamp = (uint32_t)lamp_intensity.size();
        d.lamp_pos = lamp_pos.data(); d.lamp_intensity = lamp_intensity.data(); d.lamp_distance = lamp_distance.data();
        for (int k = 0; k < 3; k++) { d.cam_pos[k] = cam_pos[k]; d.horizon[k] = horizon[k]; d.ambient[k] = ambient[k]; }
        for (int k = 0; k < 9; k++) d.cam_mat[k] = cam_mat[k];
        d.ambient_factor = ambient_factor;
        return d;
    }

    // Flat binary dump of every array (for checking the loader against another implementation):
    // per array a 16-byte name, a uint64 byte count, then the bytes.
    void dump(const std::string &path) const {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        auto put = [&](const char *name, const void *data, size_t bytes) {
            char tag[16] = {0};
            snprintf(tag, sizeof tag, "%s", name);
            uint64_t n = bytes;
            fwrite(tag, 1, 16, f); fwrite(&n, 8, 1, f);
            if (bytes) fwrite(data, 1, bytes, f);
        };
#define RT_PUT(v) put(#v, (v).data(), (v).size() * sizeof((v)[0]))
        RT_PUT(tri_p); RT_PUT(tri_e1); RT_PUT(tri_e2); RT_PUT(tri_norm); RT_PUT(tri_uv); RT_PUT(tri_mat);
        RT_PUT(sph_center); RT_PUT(sph_radius); RT_PUT(sph_mat);
        RT_PUT(mat_color); RT_PUT(mat_kd); RT_PUT(mat_ks); RT_PUT(mat_hard); RT_PUT(mat_kr); RT_PUT(mat_tex);
        RT_PUT(tex_rgba); RT_PUT(tex_offset); RT_PUT(tex_w); RT_PUT(tex_h);
        RT_PUT(lamp_pos); RT_PUT(lamp_intensity); RT_PUT(lamp_distance);
#undef RT_PUT
        put("cam_pos", cam_pos, sizeof cam_pos); put("cam_mat", cam_mat, sizeof cam_mat);
        put("horizon", horizon, sizeof horizon); put("ambient", ambient, sizeof ambient);
        fclose(f);
    }

private:
    static void vec3(const Json &a, float *out) {
        for (int k = 0; k < 3; k++) out[k] = a.at(k).f32();
    }
};

}  // namespace rt_hip
