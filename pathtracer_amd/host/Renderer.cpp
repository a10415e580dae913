// Renderer.cpp — see Renderer.h.
#include "Renderer.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace prt_host {

static void storeMat(float* dst, const mat4& m) { memcpy(dst, m.data(), 16 * sizeof(float)); }

prt_uniforms fillUniforms(const mat4& view, const mat4& projection, const mat4& lightTransform, const Settings& s,
                          uint32_t iteration, float time)
{
    prt_uniforms ubo;
    memset(&ubo, 0, sizeof ubo);
    const mat4 model(1.0f);
    mat4       proj = projection;
    proj[1][1] *= -1.0f;                                  // vkContext.cpp:1003
    const mat4 modelIT = inverseTranspose(model);         // :1004
    const mat4 pvi     = inverse(view) * inverse(proj);   // :1006
    storeMat(ubo.model, model);
    storeMat(ubo.view, view);
    storeMat(ubo.proj, proj);
    storeMat(ubo.modelIT, modelIT);
    storeMat(ubo.projViewInverse, pvi);
    storeMat(ubo.lightTransform, lightTransform);
    ubo.lightSize[0] = ubo.lightSize[1] = s.lightSourceArea;   // :1009
    ubo.lightE[0] = ubo.lightE[1] = ubo.lightE[2] = 1.0f * s.lightE;  // :1010
    ubo.lightOtherE        = s.lightOtherE;
    ubo.lightSourceArea    = s.lightSourceArea;
    ubo.numIndirectBounces = s.numIndicesBounces;
    ubo.samplesPerPixel    = s.samplesPerPixel;
    ubo.numAArays          = s.numAArays;
    ubo.filterRadius       = s.filterRadius;
    ubo.numAOrays          = s.numAOrays;
    ubo.aoRayLength        = s.aoRayLength;
    ubo.iteration          = iteration;
    ubo.time               = time;
    return ubo;
}

void Renderer::check(int rc, const char* what)
{
    if(rc != PRT_OK) throw std::runtime_error(std::string(what) + ": " + prt_last_error(m_ctx));
}

Renderer::Renderer(int device, uint32_t w, uint32_t h) : width(w), height(h)
{
    const int rc = prt_create(device, w, h, &m_ctx);
    if(rc != PRT_OK) throw std::runtime_error(std::string("prt_create: ") + prt_last_error(nullptr));
    m_camera.initDefaults((float)w / (float)h);
}

Renderer::~Renderer() { prt_destroy(m_ctx); }

void Renderer::initRaytracing(const std::vector<Model>& models)
{
    if(models.empty()) throw std::runtime_error("initRaytracing: no model");
    const Model&             m = models[0];
    std::vector<prt_texture> tex;
    for(const Texture& t : m.m_textures) tex.push_back(prt_texture{t.image.rgba.data(), t.width, t.height});
    check(prt_set_scene(m_ctx, m.m_vertices.data(), m.m_vertices.size(), m.m_indices.data(), m.m_indices.size(),
                        m.m_materials.data(), m.m_materials.size(), tex.data(), tex.size()),
          "prt_set_scene");
    check(prt_build_accel(m_ctx), "prt_build_accel");
}

prt_uniforms Renderer::updateGraphicsUniforms()
{
    // the block is filled with the light transform of the PREVIOUS frame; a pending m_moveLight captures
    // inverse(view) afterwards (vkContext.cpp:1008-1017)
    uint32_t iteration = 0;  // UniformBufferObject's default (vkContext.h:79): AO mode never overwrites it
    if(m_settings.RTX_ON)
    {
        if(m_settings.rtRenderingMode == 0 && m_settings.iteration < (uint32_t)m_settings.samplesPerPixel)
        {
            iteration = m_settings.iteration;
            m_settings.iteration += (uint32_t)m_settings.numAArays;
        }
        if(m_settings.rtRenderingMode == 1 && m_settings.iteration < (uint32_t)m_settings.numAOrays) m_settings.iteration += 1;
    }
    Settings s = m_settings;
    mat4     light = m_lightTransform;
    if(useAreaLight)
    {
        light             = m_light.getTransform();
        s.lightSourceArea = m_light.getSize().x;
        s.lightE          = m_light.getEmission().x;
    }
    prt_uniforms ubo = fillUniforms(m_camera.matrices.view, m_camera.matrices.projection, light, s, iteration, m_runTime);
    if(useAreaLight)
    {
        ubo.lightSize[1] = m_light.getSize().y;
        ubo.lightE[1]    = m_light.getEmission().y;
        ubo.lightE[2]    = m_light.getEmission().z;
    }
    if(m_moveLight)
    {
        m_lightTransform = inverse(m_camera.matrices.view);
        m_moveLight      = false;
    }
    if(m_cameraMoved)
    {
        m_cameraMoved        = false;
        m_settings.iteration = 1;
    }
    return ubo;
}

bool Renderer::renderFrame()
{
    if(m_camera.getCameraMoved())
    {
        m_cameraMoved = true;  // vkWindow sets vkContext::m_cameraMoved from the camera's flag
        m_camera.setCameraMoved();
    }
    const prt_uniforms ubo = updateGraphicsUniforms();
    if(m_settings.RTX_ON && m_settings.iteration < (uint32_t)m_settings.samplesPerPixel)
    {
        check(prt_set_uniforms(m_ctx, &ubo), "prt_set_uniforms");
        check(prt_render_iteration(m_ctx, m_settings.rtRenderingMode), "prt_render_iteration");
        return true;
    }
    return false;
}

void Renderer::render(uint32_t first, uint32_t n, uint32_t stride)
{
    if(m_moveLight)
    {
        m_lightTransform = inverse(m_camera.matrices.view);
        m_moveLight      = false;
    }
    Settings s     = m_settings;
    mat4     light = m_lightTransform;
    if(useAreaLight)
    {
        light             = m_light.getTransform();
        s.lightSourceArea = m_light.getSize().x;
        s.lightE          = m_light.getEmission().x;
    }
    prt_uniforms ubo = fillUniforms(m_camera.matrices.view, m_camera.matrices.projection, light, s, first, m_runTime);
    if(useAreaLight)
    {
        ubo.lightSize[1] = m_light.getSize().y;
        ubo.lightE[1]    = m_light.getEmission().y;
        ubo.lightE[2]    = m_light.getEmission().z;
    }
    check(prt_set_uniforms(m_ctx, &ubo), "prt_set_uniforms");
    check(prt_render(m_ctx, m_settings.rtRenderingMode, first, n, stride), "prt_render");
}

void Renderer::clear() { check(prt_clear_accum(m_ctx), "prt_clear_accum"); }

std::vector<float> Renderer::readAccum()
{
    std::vector<float> out((size_t)width * height * 4);
    check(prt_read_accum(m_ctx, out.data()), "prt_read_accum");
    return out;
}
std::vector<uint8_t> Renderer::postprocess()
{
    std::vector<uint8_t> out((size_t)width * height * 4);
    check(prt_postprocess(m_ctx, out.data()), "prt_postprocess");
    return out;
}
prt_stats Renderer::stats()
{
    prt_stats s;
    check(prt_get_stats(m_ctx, &s), "prt_get_stats");
    return s;
}

bool writePPM(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h)
{
    FILE* f = fopen(path.c_str(), "wb");
    if(!f) return false;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for(size_t i = 0; i < (size_t)w * h; ++i) fwrite(&rgba[i * 4], 1, 3, f);
    return fclose(f) == 0;
}
bool writePFM(const std::string& path, const std::vector<float>& a, uint32_t w, uint32_t h)
{
    FILE* f = fopen(path.c_str(), "wb");
    if(!f) return false;
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for(uint32_t y = h; y-- > 0;)  // PFM rows go bottom to top
        for(uint32_t x = 0; x < w; ++x)
        {
            const float* p = &a[((size_t)y * w + x) * 4];
            const float  d = p[3] != 0.0f ? p[3] : 1.0f;
            const float  c[3] = {p[0] / d, p[1] / d, p[2] / d};
            fwrite(c, sizeof(float), 3, f);
        }
    return fclose(f) == 0;
}

}  // namespace prt_host
