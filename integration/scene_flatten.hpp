// scene_flatten.hpp — flattens the reference's pointer-chasing `Scene` (scene.h:143-153) into
// the structure-of-arrays hipexec_scene_desc of include/hip_executor.h.
//
// Written as a template over the scene type so that it depends only on member NAMES
// (triangles[i]->p.x, materials[i]->color.r, camMat[i][c], ...), not on the reference's
// headers: ray_tracer_hip.cpp instantiates it with the real `Scene`, the unit test with a
// small stand-in that has the same members.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "hip_executor.h"

namespace rt_hip {

template <class SceneT>
class FlatScene {
public:
    explicit FlatScene(const SceneT &scene) { build(scene); }
    const hipexec_scene_desc &desc() const { return d_; }
    // material index of triangle / sphere i in the flattened arrays (-1 never occurs here:
    // the reference gives objects without a JSON material a fresh default Material, scene.cpp:87-89)
    int triMaterial(size_t i) const { return tri_mat_[i]; }

private:
    std::vector<float> tri_p_, tri_e1_, tri_e2_, tri_norm_, tri_uv_, sph_c_, sph_r_;
    std::vector<float> mat_color_, mat_kd_, mat_ks_, mat_hard_, mat_kr_, mat_kt_, mat_ior_, lamp_pos_, lamp_i_, lamp_d_;
    std::vector<int32_t> tri_mat_, sph_mat_, mat_tex_;
    std::vector<uint8_t> tex_rgba_;
    std::vector<uint64_t> tex_off_;
    std::vector<uint32_t> tex_w_, tex_h_;
    std::unordered_map<const void *, int32_t> mat_index_, tex_index_;
    hipexec_scene_desc d_{};

    template <class V> static void push3(std::vector<float> &dst, const V &v) { dst.push_back(v.x); dst.push_back(v.y); dst.push_back(v.z); }

    // Materials are identified by address: the ones of scene.materials first (scene.cpp:37-51),
    // then any object-private Material (the default one made for a null JSON material).
    template <class MatPtr> int32_t materialIndex(const MatPtr &m) {
        if (!m) return -1;
        auto it = mat_index_.find(static_cast<const void *>(m.get()));
        if (it != mat_index_.end()) return it->second;
        int32_t idx = (int32_t)mat_kd_.size();
        mat_index_[static_cast<const void *>(m.get())] = idx;
        mat_color_.push_back(m->color.x); mat_color_.push_back(m->color.y); mat_color_.push_back(m->color.z);
        mat_kd_.push_back(m->diffusiveFactor); mat_ks_.push_back(m->specularFactor);
        mat_hard_.push_back(m->specularHardness); mat_kr_.push_back(m->reflectionFactor);
        mat_kt_.push_back(m->refractionFactor); mat_ior_.push_back(m->refractionHardness);      // scene.h:28-29 (0 unless a caller sets them)
        int32_t tex = -1;
        if (m->textured && m->texImage) {                 // tex_image.h:13-69: RGBA8 rows top-down
            auto tt = tex_index_.find(static_cast<const void *>(m->texImage.get()));
            if (tt != tex_index_.end()) tex = tt->second;
            else {
                tex = (int32_t)tex_w_.size();
                tex_index_[static_cast<const void *>(m->texImage.get())] = tex;
                const auto *raw = m->texImage->getRawData();
                tex_off_.push_back(tex_rgba_.size());
                tex_w_.push_back(m->texImage->getWidth()); tex_h_.push_back(m->texImage->getHeight());
                tex_rgba_.insert(tex_rgba_.end(), raw->begin(), raw->end());
            }
        }
        mat_tex_.push_back(tex);
        return idx;
    }

    void build(const SceneT &s) {
        for (const auto &m : s.materials) materialIndex(m);
        for (const auto &t : s.triangles) {                // scene.h:100-120
            push3(tri_p_, t->p); push3(tri_e1_, t->e1); push3(tri_e2_, t->e2); push3(tri_norm_, t->norm);
            tri_uv_.push_back(t->uvStart.x); tri_uv_.push_back(t->uvStart.y);
            tri_uv_.push_back(t->uvU.x); tri_uv_.push_back(t->uvU.y);
            tri_uv_.push_back(t->uvV.x); tri_uv_.push_back(t->uvV.y);
            tri_mat_.push_back(materialIndex(t->material));
        }
        for (const auto &sp : s.spheres) {                 // scene.h:123-131
            push3(sph_c_, sp->center); sph_r_.push_back(sp->radius); sph_mat_.push_back(materialIndex(sp->material));
        }
        for (const auto &l : s.lamps) {                    // scene.h:134-140
            push3(lamp_pos_, l->pos); lamp_i_.push_back(l->intensity); lamp_d_.push_back(l->distance);
        }
        d_.n_tri = (uint32_t)tri_mat_.size();
        d_.tri_p = tri_p_.data(); d_.tri_e1 = tri_e1_.data(); d_.tri_e2 = tri_e2_.data(); d_.tri_norm = tri_norm_.data();
        d_.tri_uv = tri_uv_.data(); d_.tri_mat = tri_mat_.data();
        d_.n_sph = (uint32_t)sph_r_.size();
        d_.sph_center = sph_c_.data(); d_.sph_radius = sph_r_.data(); d_.sph_mat = sph_mat_.data();
        d_.n_mat = (uint32_t)mat_kd_.size();
        d_.mat_color = mat_color_.data(); d_.mat_kd = mat_kd_.data(); d_.mat_ks = mat_ks_.data();
        d_.mat_hard = mat_hard_.data(); d_.mat_kr = mat_kr_.data(); d_.mat_tex = mat_tex_.data();
        d_.mat_kt = mat_kt_.data(); d_.mat_ior = mat_ior_.data();
        d_.n_tex = (uint32_t)tex_w_.size();
        d_.tex_rgba = tex_rgba_.data(); d_.tex_offset = tex_off_.data(); d_.tex_w = tex_w_.data(); d_.tex_h = tex_h_.data();
        d_.n_lamp = (uint32_t)lamp_i_.size();
        d_.lamp_pos = lamp_pos_.data(); d_.lamp_intensity = lamp_i_.data(); d_.lamp_distance = lamp_d_.data();
        d_.cam_pos[0] = s.camPos.x; d_.cam_pos[1] = s.camPos.y; d_.cam_pos[2] = s.camPos.z;
        // glm::mat3 is column-major and scene.cpp:125-132 stores JSON transform row i as column i,
        // so camMat[i][c] = transform[i][c] = cam_mat[3*i + c]
        for (int i = 0; i < 3; i++) for (int c = 0; c < 3; c++) d_.cam_mat[3 * i + c] = s.camMat[i][c];
        d_.horizon[0] = s.worldHorizonColor.x; d_.horizon[1] = s.worldHorizonColor.y; d_.horizon[2] = s.worldHorizonColor.z;
        d_.ambient_factor = s.worldAmbientFactor;
        d_.ambient[0] = s.worldAmbientColor.x; d_.ambient[1] = s.worldAmbientColor.y; d_.ambient[2] = s.worldAmbientColor.z;
    }
};

}  // namespace rt_hip
