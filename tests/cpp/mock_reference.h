// Test-only stand-in for the TYPES the drop-in source is written against: the members of the
// reference's Scene / Triangle / Sphere / Lamp / Material / Hit / RayData / image_bitmap /
// tex_image (scene.h:21-169, image_bitmap.h:11-75, tex_image.h:13-69) and the handful of glm
// operations ray_tracer_hip.cpp itself uses.  It exists so that integration/ray_tracer_hip.cpp
// and integration/scene_flatten.hpp can be compiled and run in this repository, where the
// reference's headers (and glm) are not available.  No reference code is compiled with it.
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

namespace glm {
struct vec2 { float x, y; vec2(float a = 0, float b = 0) : x(a), y(b) {} };
struct vec3 {
    float x, y, z;
    vec3(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct mat3 {
    vec3 col[3];
    const vec3 &operator[](int i) const { return col[i]; }
    vec3 &operator[](int i) { return col[i]; }
};
inline vec3 operator*(const vec3 &v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 normalize(const vec3 &v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec3 operator*(const vec3 &v, const mat3 &m) { return vec3(dot(m[0], v), dot(m[1], v), dot(m[2], v)); }
}  // namespace glm

class tex_image {
public:
    std::vector<unsigned char> data; unsigned w = 0, h = 0;
    float scaleX = 1.0f, scaleY = 1.0f;
    std::vector<unsigned char> *getRawData() { return &data; }
    unsigned int getWidth() const { return w; }
    unsigned int getHeight() const { return h; }
    float getScaleX() const { return scaleX; }
    void setScaleX(float v) { scaleX = v; }
    float getScaleY() const { return scaleY; }
    void setScaleY(float v) { scaleY = v; }
};

#ifdef RT_USE_REFERENCE_STANDALONE_HEADERS
// Build container only (-I/root/reference): the two headers of the reference that compile on their own take the place
// of their stand-ins, so the drop-in source meets the real image_bitmap interface (getWidth / getHeight / getRawData /
// setPixel, image_bitmap.h:11-75) and the real block size (config.h:17-18).  Only the glm-dependent types below stay mocked.
#include "config.h"
#include "image_bitmap.h"
#else
#define SUB_BLOCK_WIDTH (48)
#define SUB_BLOCK_HEIGHT (48)
class image_bitmap {
    std::vector<float> d; int W, H;
public:
    image_bitmap(int w, int h) : d((size_t)w * h * 3, 0.0f), W(w), H(h) {}
    void *getRawData() const { return const_cast<float *>(d.data()); }
    void setPixel(int x, int y, float r, float g, float b) {
        if (x < 0 || x >= W || y < 0 || y >= H) return;
        float *p = d.data() + ((size_t)W * y + x) * 3; p[0] = r; p[1] = g; p[2] = b;
    }
    float *getPixel(int x, int y) const { return const_cast<float *>(d.data()) + ((size_t)W * y + x) * 3; }
    int getHeight() const { return H; }
    int getWidth() const { return W; }
    void clear() { std::fill(d.begin(), d.end(), 0.0f); }
    void copyImageTo(int x, int y, const image_bitmap &img) {               // image_bitmap.h:67-74
        for (int i = 0; i < img.getHeight() && i + y < H; i++)
            for (int j = 0; j < img.getWidth() && j + x < W; j++) {
                const float *p = img.getPixel(j, i);
                setPixel(x + j, y + i, p[0], p[1], p[2]);
            }
    }
};

#endif

class OpenClExecutor;

// members, their order and the constructor signatures follow scene.h:21-140 (every member the reference has, including
// the two refraction fields it never reads), so a drift in a name or an argument order fails to compile here too
struct Material {
    glm::vec3 color; std::shared_ptr<tex_image> texImage;
    float diffusiveFactor, specularFactor, specularHardness, reflectionFactor, refractionFactor, refractionHardness;
    bool textured;
    Material(const glm::vec3 &c = glm::vec3(0.0f, 0.0f, 0.0f), const std::shared_ptr<tex_image> &ti = std::shared_ptr<tex_image>(),
             float kd = 0.0f, float ks = 0.0f, float hard = 0.0f, float kr = 0.0f, float kt = 0.0f, float th = 0.0f, bool tex = false)
        : color(c), texImage(ti), diffusiveFactor(kd), specularFactor(ks), specularHardness(hard), reflectionFactor(kr),
          refractionFactor(kt), refractionHardness(th), textured(tex) {}
};
struct Hit {
    std::shared_ptr<Material> mtl; glm::vec3 point, norm, color; float t; bool isHit;
    Hit(bool hit, const std::shared_ptr<Material> &m = std::shared_ptr<Material>(), const glm::vec3 &p = glm::vec3(),
        const glm::vec3 &n = glm::vec3(), const glm::vec3 &c = glm::vec3(), float tt = 0.0f)
        : mtl(m), point(p), norm(n), color(c), t(tt), isHit(hit) {}
};
struct Triangle {
    glm::vec3 p, e1, e2; glm::vec2 uvStart, uvU, uvV; glm::vec3 norm; std::shared_ptr<Material> material;
    Triangle(const glm::vec3 &p_, const glm::vec3 &a, const glm::vec3 &b, const glm::vec2 &u0, const glm::vec2 &uu, const glm::vec2 &uv,
             const glm::vec3 &n, const std::shared_ptr<Material> &m) : p(p_), e1(a), e2(b), uvStart(u0), uvU(uu), uvV(uv), norm(n), material(m) {}
};
struct Sphere {
    glm::vec3 center; float radius; std::shared_ptr<Material> material;
    Sphere(const glm::vec3 &c, float r, const std::shared_ptr<Material> &m) : center(c), radius(r), material(m) {}
};
struct Lamp {
    glm::vec3 pos; float intensity, distance;
    Lamp(const glm::vec3 &p, float i, float d) : pos(p), intensity(i), distance(d) {}
};
struct Scene {
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Triangle>> triangles;
    std::vector<std::shared_ptr<Sphere>> spheres;
    std::vector<std::shared_ptr<Lamp>> lamps;
    glm::vec3 camPos; glm::mat3 camMat; glm::vec3 worldHorizonColor, worldAmbientColor; float worldAmbientFactor = 0;
};
struct RayData {
    float p_x, p_y, p_z, d_x, d_y, d_z;
    RayData(float a, float b, float c, float d, float e, float f) : p_x(a), p_y(b), p_z(c), d_x(d), d_y(e), d_z(f) {}
};

// the five declarations of ray_tracer_cl.h:11-54
void renderSceneCl(image_bitmap &outImg, const Scene &scene);
void renderSubBlockCl(image_bitmap &outImg, const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor,
                      int x, int y, int w, int h, int fullWidth, int fullHeight);
void traceRaysCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                 std::vector<glm::vec3> &outColors, int depth);
void computeClosestHitsCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                          std::vector<Hit> &hits);
void computeAnyHitsCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                      char *hits);
