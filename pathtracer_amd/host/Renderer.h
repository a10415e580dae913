// Renderer.h — headless stand-in for the part of vkContext + VkRTX that drives the ray-tracing path
// (/root/reference/src/vkContext.{h,cpp}: m_settings :161-192, updateGraphicsUniforms :993-1060, renderFrame
//  :124-231; /root/reference/src/vkRTX_setup.h:17-40). It owns a prt_context (include/prt.h) where the reference
// owns a VkRTX, keeps the Model / CameraControls / AreaLight scene API, and hands back the float4 accumulation
// image that pathRTpostprocess.comp consumes. No window, no swapchain, no UI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/prt.h"
#include "AreaLight.h"
#include "CameraControls.h"
#include "Model.h"
#include "Scene.h"

namespace prt_host {

struct Settings  // vkContext.h:161-192 (the UI's "flag system"); defaults identical
{
    bool     RTX_ON            = true;  // headless: ray tracing is the only mode (reference default: false)
    int      numIndicesBounces = 3;
    int      samplesPerPixel   = 128;
    int      numAArays         = 1;
    float    filterRadius      = 1.0f;
    int      numAOrays         = 16;
    float    aoRayLength       = 1.0f;
    float    lightSourceArea   = 0.02f;
    float    lightE            = 100.0f;
    float    lightOtherE       = 1.0f;
    int      rtRenderingMode   = 0;  // 0: Cook-Torrance path tracing, 1: AO
    uint32_t iteration         = 1;
};

// The 464-byte block exactly as vkContext::updateGraphicsUniforms fills it (without touching any state).
prt_uniforms fillUniforms(const mat4& view, const mat4& projection, const mat4& lightTransform, const Settings& s,
                          uint32_t iteration, float time);

class Renderer
{
public:
    Renderer(int device, uint32_t width, uint32_t height);  // throws std::runtime_error (VK_CHECK_RESULT semantics)
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    // VkRTX::initRaytracing(gpu, models, ubo, uboMem): models[0] only (vkRTX_setup.cpp:67, 459)
    void initRaytracing(const std::vector<Model>& models);
    // several models / instances (vkRT_TLAS.cpp:58-72 AddInstance): one BVH per model + a top level over the instances
    // on the GPU (prt_set_instances); flatten = true copies the instances into one world-space model instead (Scene.h)
    void initRaytracing(const std::vector<Model>& models, const std::vector<Instance>& instances, bool flatten = false);
    // vkRT_TLAS.cpp:178-279 with updateOnly = true: new transforms for the instances of the last initRaytracing
    void updateInstances(const std::vector<Instance>& instances);
    // vkRT_BLAS.cpp:209-290 / vkRT_TLAS.cpp:178-279 with updateOnly = true: same topology, moved vertices
    void updateVertices(const std::vector<VertexPNTC>& vertices);

    // vkContext::updateGraphicsUniforms, including its bookkeeping: the light follows the camera one frame after
    // m_moveLight was set, path mode hands out m_settings.iteration and advances it by numAArays, AO mode never
    // writes ubo.iteration (it stays 0), a camera move resets the counter after the block was filled.
    prt_uniforms updateGraphicsUniforms();
    // vkContext::renderFrame: updateGraphicsUniforms(), then one VkRTX::recordCommandBuffer while
    // iteration < samplesPerPixel. Returns whether an iteration was traced.
    bool renderFrame();
    // "N samples per pixel" without the UI loop's quirks: iterations first .. first+n-1 (stride for multi-GPU
    // sample sharding), light from m_light when useAreaLight, else the captured camera transform.
    void render(uint32_t firstIteration, uint32_t n, uint32_t stride = 1);

    // Progressive state on disk (the reference keeps m_rtRenderTarget and m_settings.iteration in memory only):
    // image + iteration counter + the settings that must match for the sum to stay meaningful. loadCheckpoint
    // throws when the image size, rendering mode, bounce count or AA ray count differ.
    void saveCheckpoint(const std::string& path);
    void loadCheckpoint(const std::string& path);

    void                 clear();
    std::vector<float>   readAccum();    // W*H*4, row-major: m_rtRenderTarget
    std::vector<uint8_t> postprocess();  // pathRTpostprocess.comp -> RGBA8
    prt_stats            stats();
    prt_context*         context() { return m_ctx; }
    const prt_uniforms&  lastUniforms() const { return m_lastUbo; }  // the block of the last renderFrame()

    Settings       m_settings;
    CameraControls m_camera;
    AreaLight      m_light;
    bool           useAreaLight     = false;  // feed ubo.lightTransform / lightSize / lightE from m_light
    bool           m_moveLight      = true;   // vkContext.h:86
    mat4           m_lightTransform = mat4(1.0f);
    bool           m_cameraMoved    = true;
    float          m_runTime        = 0.0f;
    uint32_t       width, height;

private:
    void         check(int rc, const char* what);
    prt_context* m_ctx = nullptr;
    prt_uniforms m_lastUbo{};
};

bool writePPM(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h);
bool writePFM(const std::string& path, const std::vector<float>& accum, uint32_t w, uint32_t h);  // colour / w
bool writePNG(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h);  // 8-bit RGBA

struct Checkpoint
{
    uint32_t           width = 0, height = 0;
    uint32_t           iteration = 1;  // next iteration to render (m_settings.iteration)
    int32_t            mode = 0, bounces = 0, aaRays = 1;
    std::vector<float> accum;          // W*H*4
};
bool writeCheckpoint(const std::string& path, const Checkpoint& c);
bool readCheckpoint(const std::string& path, Checkpoint& c, std::string& err);

}  // namespace prt_host
