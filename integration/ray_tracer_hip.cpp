// ray_tracer_hip.cpp — drop-in replacement of the reference's ray_tracer_cl.cpp.
//
// Defines the five functions declared in the reference's ray_tracer_cl.h (unchanged) on top
// of the C-ABI hip_executor library, so main.cpp, scene.cpp, the GLFW display and the texture
// loading stay as they are.  `class OpenClExecutor` is only forward-declared for callers
// (scene.h:19), so this file supplies its own definition: an RAII owner of a hipexec_t.
//
// Build inside the reference tree: replace ray_tracer_cl.cpp and
// opencl_executor.cpp by this file in CMakeLists.txt:26-30 and link libhip_executor.so
// instead of OpenCL.  The unit test of this repository compiles the same file against a
// stand-in for the reference's types by defining RT_HIP_REFERENCE_HEADER.
#ifdef RT_HIP_REFERENCE_HEADER
#include RT_HIP_REFERENCE_HEADER
#else
#include "ray_tracer_cl.h"
#endif

#include <algorithm>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

#include "hip_executor.h"
#include "scene_flatten.hpp"

// Error convention of the reference's executor: message on std::cerr, then
// std::runtime_error (opencl_executor.h:67-72).
static void checkHipResult(int rc, const char *what)
{
    if (rc != HIPEXEC_OK) {
        std::cerr << what << ": " << rc << " (" << hipexec_last_error() << ")" << std::endl;
        throw std::runtime_error("hip_executor function failed");
    }
}

class OpenClExecutor {
public:
    explicit OpenClExecutor(const Scene &scene) : flat_(scene)
    {
        checkHipResult(hipexec_create(&handle_, &flat_.desc(), -1), "hipexec_create");
        std::copy(flat_.desc().cam_pos, flat_.desc().cam_pos + 3, cam_pos_);
        std::copy(flat_.desc().cam_mat, flat_.desc().cam_mat + 9, cam_mat_);
    }
    ~OpenClExecutor() { hipexec_destroy(handle_); }
    OpenClExecutor(const OpenClExecutor &) = delete;
    OpenClExecutor &operator=(const OpenClExecutor &) = delete;
    hipexec_t *handle() const { return handle_; }
    // points the executor's device-side ray generation at the camera of `scene` if it is not already there
    void useCameraOf(const Scene &scene)
    {
        float pos[3] = {scene.camPos.x, scene.camPos.y, scene.camPos.z}, mat[9];
        for (int i = 0; i < 3; i++) for (int c = 0; c < 3; c++) mat[3 * i + c] = scene.camMat[i][c];
        bool same = true;
        for (int k = 0; k < 3; k++) same = same && pos[k] == cam_pos_[k];
        for (int k = 0; k < 9; k++) same = same && mat[k] == cam_mat_[k];
        if (same) return;
        checkHipResult(hipexec_set_camera(handle_, pos, mat), "hipexec_set_camera");
        std::copy(pos, pos + 3, cam_pos_); std::copy(mat, mat + 9, cam_mat_);
    }

private:
    rt_hip::FlatScene<Scene> flat_;
    hipexec_t *handle_ = nullptr;
    float cam_pos_[3], cam_mat_[9];           // the camera the executor currently renders from
};

// ray_tracer_cl.cpp:9-56.  The reference walks the frame in 48x48 blocks only to bound its
// dense result matrix; the executor renders the whole frame in one call with the same per-pixel
// arithmetic (1 sample per pixel, tone curve), straight into the image_bitmap's float RGB rows.
void renderSceneCl(image_bitmap &outImg, const Scene &scene)
{
    std::shared_ptr<OpenClExecutor> clExecutor(new OpenClExecutor(scene));
    hipexec_render_cfg cfg{};
    cfg.width = outImg.getWidth(); cfg.height = outImg.getHeight();
    cfg.mode = HIPEXEC_MODE_CL; cfg.max_depth = 0; cfg.tone = 1;
    cfg.tile_first = 0; cfg.tile_step = 1; cfg.flags = 0;
    checkHipResult(hipexec_render(clExecutor->handle(), &cfg, static_cast<float *>(outImg.getRawData()), nullptr, nullptr),
                   "hipexec_render");
}

// ray_tracer_cl.cpp:59-105.  One call: the executor makes the block's camera rays on the device from the pixel
// coordinates, traces and tones them and copies only the block back, straight into the image's rows.  The pixels are
// those renderSceneCl writes for the same frame size.  (An image whose size differs from fullWidth x fullHeight is
// served through a frame-sized staging image and setPixel, which drops what falls outside, image_bitmap.h:36-49.)
void renderSubBlockCl(image_bitmap &outImg, const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor,
                      int x, int y, int w, int h, int fullWidth, int fullHeight)
{
    // The reference makes the block's rays from the scene passed to THIS call (ray_tracer_cl.cpp:80-92); its own driver
    // always passes the scene the executor was built from.  A caller that hands over another camera gets that camera:
    // the executor's device-side ray generation is pointed at it first (the geometry stays the executor's).
    clExecutor->useCameraOf(scene);
    hipexec_render_cfg cfg{};
    cfg.width = fullWidth; cfg.height = fullHeight;
    cfg.mode = HIPEXEC_MODE_CL; cfg.max_depth = 0; cfg.tone = 1;
    cfg.tile_first = 0; cfg.tile_step = 1; cfg.flags = 0;
    if (outImg.getWidth() == fullWidth && outImg.getHeight() == fullHeight) {
        checkHipResult(hipexec_render_rect(clExecutor->handle(), &cfg, x, y, w, h, static_cast<float *>(outImg.getRawData()),
                                           nullptr, nullptr), "hipexec_render_rect");
        return;
    }
    std::vector<float> frame((size_t)fullWidth * (size_t)fullHeight * 3);
    checkHipResult(hipexec_render_rect(clExecutor->handle(), &cfg, x, y, w, h, frame.data(), nullptr, nullptr), "hipexec_render_rect");
    for (int py = std::max(0, y); py < std::min(fullHeight, y + h); py++)
        for (int px = std::max(0, x); px < std::min(fullWidth, x + w); px++) {
            const float *c = &frame[((size_t)py * fullWidth + px) * 3];
            outImg.setPixel(px, py, c[0], c[1], c[2]);
        }
}

// ray_tracer_cl.cpp:108-178: outColors[i] is accumulated into; depth is accepted and ignored (:114)
void traceRaysCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                 std::vector<glm::vec3> &outColors, int depth)
{
    (void)scene; (void)depth;
    if (rays.empty()) return;
    static_assert(sizeof(RayData) == 6 * sizeof(float), "RayData must pack 6 floats (scene.h:156-169)");
    std::vector<float> colours(3 * rays.size());
    for (size_t i = 0; i < rays.size(); i++) { colours[3 * i] = outColors[i].x; colours[3 * i + 1] = outColors[i].y; colours[3 * i + 2] = outColors[i].z; }
    checkHipResult(hipexec_trace_rays(clExecutor->handle(), HIPEXEC_MODE_CL, 0, reinterpret_cast<const float *>(rays.data()),
                                      (uint32_t)rays.size(), colours.data(), nullptr), "hipexec_trace_rays");
    for (size_t i = 0; i < rays.size(); i++) outColors[i] = glm::vec3(colours[3 * i], colours[3 * i + 1], colours[3 * i + 2]);
}

// ray_tracer_cl.cpp:181-242: hits is cleared and resized; misses keep Hit(false)
void computeClosestHitsCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                          std::vector<Hit> &hits)
{
    hits.clear();
    hits.resize(rays.size(), Hit(false));
    if (rays.empty()) return;
    std::vector<hipexec_hit> out(rays.size());
    checkHipResult(hipexec_closest_hits(clExecutor->handle(), HIPEXEC_MODE_CL, reinterpret_cast<const float *>(rays.data()),
                                        (uint32_t)rays.size(), out.data()), "hipexec_closest_hits");
    const size_t nTri = scene.triangles.size();
    for (size_t i = 0; i < rays.size(); i++) {
        const hipexec_hit &h = out[i];
        if (h.id < 0) continue;
        hits[i].isHit = true;
        hits[i].mtl = ((size_t)h.id < nTri) ? scene.triangles[(size_t)h.id]->material : scene.spheres[(size_t)h.id - nTri]->material;
        hits[i].point = glm::vec3(h.point[0], h.point[1], h.point[2]);
        hits[i].norm = glm::vec3(h.norm[0], h.norm[1], h.norm[2]);
        hits[i].color = glm::vec3(h.color[0], h.color[1], h.color[2]);
        hits[i].t = h.t;
    }
}

// ray_tracer_cl.cpp:245-262: hits[i] = hits[i] || occluded
void computeAnyHitsCl(const Scene &scene, std::shared_ptr<OpenClExecutor> clExecutor, const std::vector<RayData> &rays,
                      char *hits)
{
    (void)scene;
    if (rays.empty()) return;
    checkHipResult(hipexec_any_hits(clExecutor->handle(), HIPEXEC_MODE_CL, reinterpret_cast<const float *>(rays.data()),
                                    (uint32_t)rays.size(), hits), "hipexec_any_hits");
}

#ifdef RT_HIP_REFERENCE_HEADER
// test hook: the reference constructs executors only inside renderSceneCl (ray_tracer_cl.cpp:14)
std::shared_ptr<OpenClExecutor> makeExecutorForTest(const Scene &scene)
{
    return std::shared_ptr<OpenClExecutor>(new OpenClExecutor(scene));
}
#endif
