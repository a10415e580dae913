// host_capi.cpp — include/prt_host.h over the C++ host layer.
#include "../../include/prt_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "Model.h"
#include "Renderer.h"
#include "obj_loader.h"

using namespace prt_host;

struct prth_model
{
    Model m;
};

static void set_err(char* err, size_t n, const std::string& s)
{
    if(err && n)
    {
        strncpy(err, s.c_str(), n - 1);
        err[n - 1] = '\0';
    }
}

static Settings to_settings(const prth_settings* s)
{
    Settings o;
    if(s)
    {
        o.numIndicesBounces = s->numIndicesBounces;
        o.samplesPerPixel   = s->samplesPerPixel;
        o.numAArays         = s->numAArays;
        o.filterRadius      = s->filterRadius;
        o.numAOrays         = s->numAOrays;
        o.aoRayLength       = s->aoRayLength;
        o.lightSourceArea   = s->lightSourceArea;
        o.lightE            = s->lightE;
        o.lightOtherE       = s->lightOtherE;
        o.rtRenderingMode   = s->rtRenderingMode;
        o.iteration         = s->iteration;
    }
    return o;
}

extern "C" {

int prth_model_load(const char* path, prth_model** out, char* err, size_t errLen)
{
    if(!path || !out) return -1;
    try
    {
        prth_model* m = new prth_model{Model(std::string(path))};
        *out          = m;
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}
void   prth_model_free(prth_model* m) { delete m; }
size_t prth_model_num_vertices(const prth_model* m) { return m->m.m_vertices.size(); }
size_t prth_model_num_indices(const prth_model* m) { return m->m.m_indices.size(); }
size_t prth_model_num_materials(const prth_model* m) { return m->m.m_materials.size(); }
size_t prth_model_num_textures(const prth_model* m) { return m->m.m_textures.size(); }
const prt_vertex*   prth_model_vertices(const prth_model* m) { return m->m.m_vertices.data(); }
const uint32_t*     prth_model_indices(const prth_model* m) { return m->m.m_indices.data(); }
const prt_material* prth_model_materials(const prth_model* m) { return m->m.m_materials.data(); }
int prth_model_texture(const prth_model* m, size_t i, prt_texture* out, const char** path)
{
    if(!m || i >= m->m.m_textures.size() || !out) return -1;
    const Texture& t = m->m.m_textures[i];
    out->rgba        = t.image.rgba.data();
    out->width       = t.width;
    out->height      = t.height;
    if(path) *path = t.path.c_str();
    return 0;
}
const char* prth_model_warnings(const prth_model* m) { return m->m.warnings.c_str(); }

static std::string jesc(const std::string& s)
{
    std::string o;
    for(char c : s)
    {
        if(c == '\n') { o += "\\n"; continue; }
        if(c == '\t') { o += "\\t"; continue; }
        if(c == '\\' || c == '"') o += '\\';
        o += c;
    }
    return o;
}
int prth_obj_dump_json(const char* objPath, const char* mtlDir, const char* jsonPath)
{
    ObjData     d;
    std::string err;
    const bool  ok = load_obj(objPath, mtlDir ? mtlDir : "", d, err);
    FILE*       f  = fopen(jsonPath, "w");
    if(!f) return -1;
    auto farr = [&](const char* n, const std::vector<float>& v) {
        fprintf(f, "\"%s\": [", n);
        for(size_t i = 0; i < v.size(); ++i) fprintf(f, "%.9g%s", v[i], i + 1 < v.size() ? ", " : "");
        fprintf(f, "],\n");
    };
    fprintf(f, "{\"ok\": %s,\n", ok ? "true" : "false");
    farr("vertices", d.vertices);
    farr("normals", d.normals);
    farr("texcoords", d.texcoords);
    farr("colors", d.colors);
    fprintf(f, "\"shapes\": [\n");
    for(size_t s = 0; s < d.shapes.size(); ++s)
    {
        fprintf(f, "{\"name\": \"%s\", \"indices\": [", jesc(d.shapes[s].name).c_str());
        const auto& ix = d.shapes[s].indices;
        for(size_t i = 0; i < ix.size(); ++i)
            fprintf(f, "[%d, %d, %d]%s", ix[i].vertex_index, ix[i].normal_index, ix[i].texcoord_index, i + 1 < ix.size() ? ", " : "");
        fprintf(f, "], \"material_ids\": [");
        const auto& mi = d.shapes[s].material_ids;
        for(size_t i = 0; i < mi.size(); ++i) fprintf(f, "%d%s", mi[i], i + 1 < mi.size() ? ", " : "");
        fprintf(f, "]}%s\n", s + 1 < d.shapes.size() ? "," : "");
    }
    fprintf(f, "],\n\"materials\": [\n");
    for(size_t m = 0; m < d.materials.size(); ++m)
    {
        const ObjMaterial& t = d.materials[m];
        fprintf(f,
                "{\"name\": \"%s\", \"ambient\": [%.9g, %.9g, %.9g], \"diffuse\": [%.9g, %.9g, %.9g], "
                "\"specular\": [%.9g, %.9g, %.9g], \"transmittance\": [%.9g, %.9g, %.9g], "
                "\"emission\": [%.9g, %.9g, %.9g], \"shininess\": %.9g, \"ior\": %.9g, \"dissolve\": %.9g, "
                "\"illum\": %d, \"roughness\": %.9g, \"metallic\": %.9g, \"diffuse_texname\": \"%s\", "
                "\"specular_texname\": \"%s\", \"bump_texname\": \"%s\"}%s\n",
                jesc(t.name).c_str(), t.ambient[0], t.ambient[1], t.ambient[2], t.diffuse[0], t.diffuse[1], t.diffuse[2],
                t.specular[0], t.specular[1], t.specular[2], t.transmittance[0], t.transmittance[1], t.transmittance[2],
                t.emission[0], t.emission[1], t.emission[2], t.shininess, t.ior, t.dissolve, t.illum, t.roughness, t.metallic,
                jesc(t.diffuse_texname).c_str(), jesc(t.specular_texname).c_str(), jesc(t.bump_texname).c_str(),
                m + 1 < d.materials.size() ? "," : "");
    }
    fprintf(f, "], \"warn\": \"%s\", \"err\": \"%s\"}\n", jesc(d.warn).c_str(), jesc(err).c_str());
    fclose(f);
    return ok ? 0 : -2;
}

int prth_image_load(const char* path, uint32_t* w, uint32_t* h, uint8_t** rgba)
{
    if(!path || !w || !h || !rgba) return -1;
    Image img;
    if(!load_image(path, img)) return -2;
    *w    = img.width;
    *h    = img.height;
    *rgba = (uint8_t*)malloc(img.rgba.size());
    if(!*rgba) return -3;
    memcpy(*rgba, img.rgba.data(), img.rgba.size());
    return 0;
}
void prth_free(void* p) { free(p); }

void prth_default_settings(prth_settings* s)
{
    const Settings d;
    s->numIndicesBounces = d.numIndicesBounces;
    s->samplesPerPixel   = d.samplesPerPixel;
    s->numAArays         = d.numAArays;
    s->filterRadius      = d.filterRadius;
    s->numAOrays         = d.numAOrays;
    s->aoRayLength       = d.aoRayLength;
    s->lightSourceArea   = d.lightSourceArea;
    s->lightE            = d.lightE;
    s->lightOtherE       = d.lightOtherE;
    s->rtRenderingMode   = d.rtRenderingMode;
    s->iteration         = d.iteration;
}

int prth_camera_uniforms(float fov, float aspect, float zn, float zf, const float pos[3], float yaw, float pitch,
                         const prth_settings* settings, prt_uniforms* ubo, float projOut[16])
{
    if(!pos || !ubo) return -1;
    CameraControls cam;
    cam.m_fov  = fov;
    cam.m_near = zn;
    cam.m_far  = zf;
    cam.setPosition(vec3(pos[0], pos[1], pos[2]));
    cam.initDefaults(aspect);
    cam.setRotation(yaw, pitch);
    const Settings s     = to_settings(settings);
    const mat4     light = inverse(cam.matrices.view);  // m_moveLight: the light sits at the camera
    *ubo                 = fillUniforms(cam.matrices.view, cam.matrices.projection, light, s, s.iteration, 0.0f);
    if(projOut) memcpy(projOut, cam.matrices.projection.data(), 64);
    return 0;
}

static void setup_view(Renderer& r, const prth_settings* settings, uint32_t w, uint32_t h, const float eye[3],
                       const float target[3], const float* lightPos, const float* lightDir)
{
    r.m_settings = to_settings(settings);
    r.m_camera.setPosition(vec3(eye[0], eye[1], eye[2]));
    r.m_camera.initDefaults((float)w / (float)h);
    r.m_camera.lookAtPoint(vec3(target[0], target[1], target[2]));
    if(lightPos && lightDir)
    {
        // an AreaLight at lightPos whose normal (-Z axis) is lightDir, built like a look-at frame
        const vec3 d  = normalize(vec3(lightDir[0], lightDir[1], lightDir[2]));
        const vec3 up = std::fabs(d.y) < 0.99f ? vec3(0, 1, 0) : vec3(0, 0, -1);
        mat3       R;
        R[2] = -d;
        R[0] = normalize(cross(up, R[2]));
        R[1] = cross(R[2], R[0]);
        r.m_light.setOrientation(R);
        r.m_light.setPosition(vec3(lightPos[0], lightPos[1], lightPos[2]));
        r.m_light.setSize(vec2{r.m_settings.lightSourceArea, r.m_settings.lightSourceArea});
        r.m_light.setEmission(vec3(r.m_settings.lightE, r.m_settings.lightE, r.m_settings.lightE));
        r.useAreaLight = true;
    }
}

int prth_render_obj(const char* objPath, int device, uint32_t w, uint32_t h, const prth_settings* settings, const float eye[3],
                    const float target[3], const float* lightPos, const float* lightDir, uint32_t first, uint32_t n,
                    float* accumOut, prt_uniforms* uboOut, char* err, size_t errLen)
{
    if(!objPath || !eye || !target || !accumOut) return -1;
    try
    {
        std::vector<Model> models;
        models.emplace_back(std::string(objPath));
        Renderer r(device, w, h);
        setup_view(r, settings, w, h, eye, target, lightPos, lightDir);
        r.initRaytracing(models);
        if(uboOut)
        {
            Settings s     = r.m_settings;
            mat4     light = r.useAreaLight ? r.m_light.getTransform() : inverse(r.m_camera.matrices.view);
            *uboOut        = fillUniforms(r.m_camera.matrices.view, r.m_camera.matrices.projection, light, s, first, 0.0f);
        }
        r.render(first, n, 1);
        const std::vector<float> a = r.readAccum();
        memcpy(accumOut, a.data(), a.size() * sizeof(float));
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

int prth_flatten_instances(const prth_model* const* models, size_t nModels, const prth_instance* instances, size_t nInst,
                           prth_model** out, uint32_t* firstTriangle, char* err, size_t errLen)
{
    if(!models || !instances || !out) return -1;
    try
    {
        std::vector<Model> ms;
        for(size_t i = 0; i < nModels; ++i)
        {
            if(!models[i]) return -1;
            ms.push_back(models[i]->m);
        }
        std::vector<Instance> is(nInst);
        for(size_t i = 0; i < nInst; ++i)
        {
            is[i].modelIndex = instances[i].modelIndex;
            memcpy(&is[i].transform[0][0], instances[i].transform, 64);
            is[i].instanceId    = instances[i].instanceId;
            is[i].hitGroupIndex = instances[i].hitGroupIndex;
        }
        FlatScene fs = flattenInstances(ms, is);
        if(firstTriangle) memcpy(firstTriangle, fs.firstTriangle.data(), fs.firstTriangle.size() * sizeof(uint32_t));
        *out = new prth_model{std::move(fs.model)};
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

static void to_scene(const prth_model* const* models, size_t nModels, const prth_instance* instances, size_t nInst,
                     std::vector<Model>& ms, std::vector<Instance>& is)
{
    for(size_t i = 0; i < nModels; ++i)
    {
        if(!models[i]) throw std::runtime_error("null model");
        ms.push_back(models[i]->m);
    }
    is.resize(nInst);
    for(size_t i = 0; i < nInst; ++i)
    {
        is[i].modelIndex = instances[i].modelIndex;
        memcpy(&is[i].transform[0][0], instances[i].transform, 64);
        is[i].instanceId    = instances[i].instanceId;
        is[i].hitGroupIndex = instances[i].hitGroupIndex;
    }
}

int prth_pool_instances(const prth_model* const* models, size_t nModels, const prth_instance* instances, size_t nInst,
                        prth_model** out, prt_blas_range* ranges, prt_instance* instOut, char* err, size_t errLen)
{
    if(!models || !instances || !out) return -1;
    try
    {
        std::vector<Model>    ms;
        std::vector<Instance> is;
        to_scene(models, nModels, instances, nInst, ms, is);
        PooledScene ps = poolInstances(ms, is);
        if(ranges) memcpy(ranges, ps.ranges.data(), ps.ranges.size() * sizeof(prt_blas_range));
        if(instOut) memcpy(instOut, ps.instances.data(), ps.instances.size() * sizeof(prt_instance));
        *out = new prth_model{std::move(ps.model)};
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

int prth_render_instances(const prth_model* const* models, size_t nModels, const prth_instance* instances, size_t nInst,
                          int flatten, int device, uint32_t w, uint32_t h, const prth_settings* settings, const float eye[3],
                          const float target[3], const float* lightPos, const float* lightDir, uint32_t first, uint32_t n,
                          const prth_instance* moved, float* accumOut, uint32_t* hitIdsOut, char* err, size_t errLen)
{
    if(!models || !instances || !eye || !target || !accumOut) return -1;
    try
    {
        std::vector<Model>    ms;
        std::vector<Instance> is;
        to_scene(models, nModels, instances, nInst, ms, is);
        Renderer r(device, w, h);
        setup_view(r, settings, w, h, eye, target, lightPos, lightDir);
        r.initRaytracing(ms, is, flatten != 0);
        if(moved)
        {
            std::vector<Model>    dummy;
            std::vector<Instance> is2;
            to_scene(models, 0, moved, nInst, dummy, is2);
            if(flatten)
                r.initRaytracing(ms, is2, true);
            else
                r.updateInstances(is2);
        }
        r.render(first, n, 1);
        const std::vector<float> a = r.readAccum();
        memcpy(accumOut, a.data(), a.size() * sizeof(float));
        if(hitIdsOut && prt_read_hit_ids(r.context(), hitIdsOut) != PRT_OK) throw std::runtime_error(prt_last_error(r.context()));
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

int prth_render_model(const prth_model* model, int device, uint32_t w, uint32_t h, const prth_settings* settings,
                      const float eye[3], const float target[3], const float* lightPos, const float* lightDir, uint32_t first,
                      uint32_t n, const char* resumeFrom, const char* saveTo, float* accumOut, char* err, size_t errLen)
{
    if(!model || !eye || !target || !accumOut) return -1;
    try
    {
        const std::vector<Model> models(1, model->m);
        Renderer                 r(device, w, h);
        setup_view(r, settings, w, h, eye, target, lightPos, lightDir);
        r.initRaytracing(models);
        if(resumeFrom && *resumeFrom)
        {
            r.loadCheckpoint(resumeFrom);
            first = r.m_settings.iteration;
        }
        r.render(first, n, 1);
        r.m_settings.iteration = first + n;  // Renderer::render hands out first, first+1, ...
        if(saveTo && *saveTo) r.saveCheckpoint(saveTo);
        const std::vector<float> a = r.readAccum();
        memcpy(accumOut, a.data(), a.size() * sizeof(float));
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

int prth_frame_loop(const prth_model* model, int device, uint32_t w, uint32_t h, const prth_settings* settings,
                    const float eye[3], const float target[3], const prth_frame_event* events, size_t nEvents, uint32_t nFrames,
                    prth_frame_record* records, float* accumOut, char* err, size_t errLen)
{
    if(!model || !eye || !target || !records || (nEvents && !events)) return -1;
    try
    {
        const std::vector<Model> models(1, model->m);
        Renderer                 r(device, w, h);
        setup_view(r, settings, w, h, eye, target, nullptr, nullptr);
        r.initRaytracing(models);
        for(uint32_t f = 0; f < nFrames; ++f)
        {
            // the window's input handling runs before renderFrame (vkContext::mainLoop, vkContext.cpp:92-118)
            for(size_t k = 0; k < nEvents; ++k)
            {
                const prth_frame_event& e = events[k];
                if(e.frame != f) continue;
                if(e.kind == PRTH_EVENT_MOVE_CAMERA)
                    r.m_camera.updateMovements(1.0f, vec3(e.v[0], e.v[1], e.v[2]));  // WASD: vkWindow.cpp:127-134
                else if(e.kind == PRTH_EVENT_MOVE_LIGHT)
                    r.m_moveLight = true;  // Space: vkContext.cpp:1680-1705
                else if(e.kind == PRTH_EVENT_SET_MODE)
                    r.m_settings.rtRenderingMode = (int)e.v[0];  // KP1 / KP2
                else
                    throw std::runtime_error("prth_frame_loop: unknown event kind");
            }
            prth_frame_record& rec = records[f];
            const uint32_t     before = r.m_settings.iteration;
            rec.traced                = r.renderFrame() ? 1 : 0;
            rec.ubo                   = r.lastUniforms();
            rec.settings_iteration_before = before;
            rec.settings_iteration_after  = r.m_settings.iteration;
        }
        if(accumOut)
        {
            const std::vector<float> a = r.readAccum();
            memcpy(accumOut, a.data(), a.size() * sizeof(float));
        }
        return 0;
    }
    catch(const std::exception& e)
    {
        set_err(err, errLen, e.what());
        return -2;
    }
}

int prth_write_png(const char* path, const uint8_t* rgba, uint32_t w, uint32_t h)
{
    if(!path || !rgba) return -1;
    const std::vector<uint8_t> v(rgba, rgba + (size_t)w * h * 4);
    return writePNG(path, v, w, h) ? 0 : -2;
}

int prth_checkpoint_write(const char* path, uint32_t w, uint32_t h, uint32_t nextIteration, int32_t mode, int32_t bounces,
                          int32_t aaRays, const float* accum)
{
    if(!path || !accum) return -1;
    Checkpoint c;
    c.width     = w;
    c.height    = h;
    c.iteration = nextIteration;
    c.mode      = mode;
    c.bounces   = bounces;
    c.aaRays    = aaRays;
    c.accum.assign(accum, accum + (size_t)w * h * 4);
    return writeCheckpoint(path, c) ? 0 : -2;
}

int prth_checkpoint_read(const char* path, uint32_t* w, uint32_t* h, uint32_t* nextIteration, int32_t* mode, int32_t* bounces,
                         int32_t* aaRays, float** accum, char* err, size_t errLen)
{
    if(!path || !w || !h || !accum) return -1;
    Checkpoint  c;
    std::string e;
    if(!readCheckpoint(path, c, e))
    {
        set_err(err, errLen, e);
        return -2;
    }
    *w = c.width;
    *h = c.height;
    if(nextIteration) *nextIteration = c.iteration;
    if(mode) *mode = c.mode;
    if(bounces) *bounces = c.bounces;
    if(aaRays) *aaRays = c.aaRays;
    *accum = (float*)malloc(c.accum.size() * sizeof(float));
    if(!*accum) return -3;
    memcpy(*accum, c.accum.data(), c.accum.size() * sizeof(float));
    return 0;
}

}  // extern "C"
