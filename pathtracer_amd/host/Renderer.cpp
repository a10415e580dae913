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

void Renderer::initRaytracing(const std::vector<Model>& models, const std::vector<Instance>& instances, bool flatten)
{
    if(flatten)
    {
        const FlatScene          fs = flattenInstances(models, instances);
        const std::vector<Model> one(1, fs.model);
        initRaytracing(one);
        return;
    }
    const PooledScene        ps = poolInstances(models, instances);
    const Model&             m  = ps.model;
    std::vector<prt_texture> tex;
    for(const Texture& t : m.m_textures) tex.push_back(prt_texture{t.image.rgba.data(), t.width, t.height});
    check(prt_set_scene(m_ctx, m.m_vertices.data(), m.m_vertices.size(), m.m_indices.data(), m.m_indices.size(),
                        m.m_materials.data(), m.m_materials.size(), tex.data(), tex.size()),
          "prt_set_scene");
    check(prt_set_instances(m_ctx, ps.ranges.data(), ps.ranges.size(), ps.instances.data(), ps.instances.size()),
          "prt_set_instances");
    check(prt_build_accel(m_ctx), "prt_build_accel");
}

void Renderer::updateInstances(const std::vector<Instance>& instances)
{
    std::vector<float> xf(instances.size() * 16);
    for(size_t i = 0; i < instances.size(); ++i) memcpy(&xf[16 * i], instances[i].transform.data(), 64);
    check(prt_update_instance_transforms(m_ctx, xf.data(), instances.size()), "prt_update_instance_transforms");
}

void Renderer::updateVertices(const std::vector<VertexPNTC>& vertices)
{
    check(prt_update_vertices(m_ctx, vertices.data(), vertices.size()), "prt_update_vertices");
    check(prt_refit_accel(m_ctx), "prt_refit_accel");
}

void Renderer::saveCheckpoint(const std::string& path)
{
    Checkpoint c;
    c.width     = width;
    c.height    = height;
    c.iteration = m_settings.iteration;
    c.mode      = m_settings.rtRenderingMode;
    c.bounces   = m_settings.numIndicesBounces;
    c.aaRays    = m_settings.numAArays;
    c.accum     = readAccum();
    if(!writeCheckpoint(path, c)) throw std::runtime_error("saveCheckpoint: cannot write " + path);
}

void Renderer::loadCheckpoint(const std::string& path)
{
    Checkpoint  c;
    std::string err;
    if(!readCheckpoint(path, c, err)) throw std::runtime_error("loadCheckpoint: " + err);
    if(c.width != width || c.height != height)
        throw std::runtime_error("loadCheckpoint: image is " + std::to_string(c.width) + "x" + std::to_string(c.height) +
                                 ", renderer is " + std::to_string(width) + "x" + std::to_string(height));
    if(c.mode != m_settings.rtRenderingMode || c.bounces != m_settings.numIndicesBounces || c.aaRays != m_settings.numAArays)
        throw std::runtime_error("loadCheckpoint: rendering mode / bounces / AA rays differ from the saved render");
    check(prt_write_accum(m_ctx, c.accum.data()), "prt_write_accum");
    m_settings.iteration = c.iteration;
    m_cameraMoved        = false;  // or the next frame would restart the accumulation
    m_camera.setCameraMoved();
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
    m_lastUbo              = ubo;
    if(m_settings.RTX_ON && m_settings.iteration < (uint32_t)m_settings.samplesPerPixel)
    {
        // the frame loop has the reference's image semantics: the shader restarts the sum when ubo.iteration <= 1
        // (pathRT.rgen:649-654; AO.rgen likewise, and AO never gets past 0) and a primary miss overwrites the pixel
        check(prt_set_flags(m_ctx, PRT_FLAG_LITERAL), "prt_set_flags");
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
    check(prt_set_flags(m_ctx, 0), "prt_set_flags");  // additive sums: what sample sharding over GPUs needs
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


// ---- PNG (8-bit RGBA, filter 0, stored deflate blocks: no compression, any decoder reads it) -------------------
static uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc)
{
    static uint32_t table[256];
    if(!table[1])
        for(uint32_t i = 0; i < 256; ++i)
        {
            uint32_t c = i;
            for(int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    crc = ~crc;
    for(size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}
static void put32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
static bool writeChunk(FILE* f, const char type[4], const std::vector<uint8_t>& data)
{
    std::vector<uint8_t> head;
    put32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t crc = crc32_of((const uint8_t*)type, 4, 0);
    crc          = crc32_of(data.data(), data.size(), crc);
    std::vector<uint8_t> tail;
    put32(tail, crc);
    return fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           fwrite(tail.data(), 1, 4, f) == 4;
}
bool writePNG(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h)
{
    if(w == 0 || h == 0 || rgba.size() < (size_t)w * h * 4) return false;
    FILE* f = fopen(path.c_str(), "wb");
    if(!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    bool                 ok     = fwrite(sig, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr;
    put32(ihdr, w);
    put32(ihdr, h);
    const uint8_t rest[5] = {8, 6, 0, 0, 0};  // bit depth 8, colour type 6 (RGBA), deflate, adaptive, no interlace
    ihdr.insert(ihdr.end(), rest, rest + 5);
    ok = ok && writeChunk(f, "IHDR", ihdr);
    // raw scanlines, each with filter byte 0
    const size_t         stride = (size_t)w * 4 + 1;
    std::vector<uint8_t> raw(stride * h);
    for(uint32_t y = 0; y < h; ++y)
    {
        raw[y * stride] = 0;
        memcpy(&raw[y * stride + 1], &rgba[(size_t)y * w * 4], (size_t)w * 4);
    }
    uint32_t a = 1, b = 0;  // adler32
    for(uint8_t c : raw)
    {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    for(size_t pos = 0; pos < raw.size();)
    {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff));
        z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    }
    put32(z, (b << 16) | a);
    ok = ok && writeChunk(f, "IDAT", z);
    ok = ok && writeChunk(f, "IEND", {});
    return (fclose(f) == 0) && ok;
}

// ---- checkpoint file: 64-byte little-endian header, then W*H float4 ------------------------------------------------
namespace {
struct CkptHeader
{
    char     magic[8];  // "PRTCKPT1"
    uint32_t width, height, iteration;
    int32_t  mode, bounces, aaRays;
    uint64_t payloadBytes;
    uint32_t payloadCrc;
    uint32_t reserved[5];
};
static_assert(sizeof(CkptHeader) == 64, "checkpoint header is 64 bytes");
}  // namespace

bool writeCheckpoint(const std::string& path, const Checkpoint& c)
{
    if(c.accum.size() != (size_t)c.width * c.height * 4) return false;
    CkptHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "PRTCKPT1", 8);
    h.width        = c.width;
    h.height       = c.height;
    h.iteration    = c.iteration;
    h.mode         = c.mode;
    h.bounces      = c.bounces;
    h.aaRays       = c.aaRays;
    h.payloadBytes = c.accum.size() * sizeof(float);
    h.payloadCrc   = crc32_of((const uint8_t*)c.accum.data(), (size_t)h.payloadBytes, 0);
    FILE* f        = fopen(path.c_str(), "wb");
    if(!f) return false;
    const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(c.accum.data(), 1, (size_t)h.payloadBytes, f) == h.payloadBytes;
    return (fclose(f) == 0) && ok;
}

bool readCheckpoint(const std::string& path, Checkpoint& c, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if(!f)
    {
        err = "cannot open " + path;
        return false;
    }
    CkptHeader h;
    bool       ok = fread(&h, sizeof h, 1, f) == 1;
    if(!ok || memcmp(h.magic, "PRTCKPT1", 8) != 0)
    {
        fclose(f);
        err = path + " is not a checkpoint file";
        return false;
    }
    if(h.width == 0 || h.height == 0 || h.payloadBytes != (uint64_t)h.width * h.height * 16)
    {
        fclose(f);
        err = path + ": inconsistent header";
        return false;
    }
    c.width     = h.width;
    c.height    = h.height;
    c.iteration = h.iteration;
    c.mode      = h.mode;
    c.bounces   = h.bounces;
    c.aaRays    = h.aaRays;
    c.accum.resize((size_t)h.width * h.height * 4);
    ok = fread(c.accum.data(), 1, (size_t)h.payloadBytes, f) == h.payloadBytes;
    uint8_t extra;
    const bool trailing = fread(&extra, 1, 1, f) == 1;
    fclose(f);
    if(!ok || trailing)
    {
        err = path + (ok ? ": trailing bytes" : ": truncated");
        return false;
    }
    if(crc32_of((const uint8_t*)c.accum.data(), (size_t)h.payloadBytes, 0) != h.payloadCrc)
    {
        err = path + ": checksum mismatch";
        return false;
    }
    return true;
}

}  // namespace prt_host
