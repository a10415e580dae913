// TEST INFRASTRUCTURE (fixture generator, runs only in the build container where /root/reference is mounted).
// Uses the header-only libraries vendored in the reference (glm 0.9.9.5, tinyobjloader, stb_image) exactly the
// way the reference's host code calls them, and writes golden vectors for the headless host layer:
//   camera   : glm::perspective / glm::lookAt / inverse as CameraControls.cpp:35-41,111-126 and
//              vkContext.cpp:1000-1006 use them (GLM_FORCE_RADIANS, GLM_FORCE_DEPTH_ZERO_TO_ONE)
//   obj      : tinyobj::LoadObj output (Model.cpp:54) for tests/golden/obj/mixed.obj
//   textures : stbi_load(..., STBI_rgb_alpha) output (Model.cpp:350)
// usage: ref_host_dump <obj> <mtl dir> <out.json> <texture files...>
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_ENABLE_EXPERIMENTAL
#include <glm/glm.hpp>
#include <glm/gtc/matrix_inverse.hpp>
#include <glm/gtc/matrix_transform.hpp>
#define TINYOBJLOADER_IMPLEMENTATION
#include <tinyobjloader/tiny_obj_loader.h>
#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

static std::string esc(const std::string& s)
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

static void mat(FILE* f, const char* name, const glm::mat4& m, bool comma = true)
{
    fprintf(f, "\"%s\": [", name);
    const float* p = &m[0][0];
    for(int i = 0; i < 16; ++i) fprintf(f, "%.9g%s", p[i], i < 15 ? ", " : "");
    fprintf(f, "]%s\n", comma ? "," : "");
}

// tinyobj::LoadObj as Model.cpp:54 calls it (triangulate = true, default_vcols_fallback = true), dumped as one JSON object
static void dump_obj(FILE* f, const char* objPath, const char* mtlDir)
{
    tinyobj::attrib_t                attrib;
    std::vector<tinyobj::shape_t>    shapes;
    std::vector<tinyobj::material_t> materials;
    std::string                      warn, err;
    bool ok = tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, objPath, mtlDir);
    fprintf(f, "\"obj\": {\"ok\": %s,\n", ok ? "true" : "false");
    auto farr = [&](const char* n, const std::vector<float>& v) {
        fprintf(f, "\"%s\": [", n);
        for(size_t i = 0; i < v.size(); ++i) fprintf(f, "%.9g%s", v[i], i + 1 < v.size() ? ", " : "");
        fprintf(f, "],\n");
    };
    farr("vertices", attrib.vertices);
    farr("normals", attrib.normals);
    farr("texcoords", attrib.texcoords);
    farr("colors", attrib.colors);
    fprintf(f, "\"shapes\": [\n");
    for(size_t s = 0; s < shapes.size(); ++s)
    {
        fprintf(f, "{\"name\": \"%s\", \"indices\": [", esc(shapes[s].name).c_str());
        const auto& ix = shapes[s].mesh.indices;
        for(size_t i = 0; i < ix.size(); ++i)
            fprintf(f, "[%d, %d, %d]%s", ix[i].vertex_index, ix[i].normal_index, ix[i].texcoord_index,
                    i + 1 < ix.size() ? ", " : "");
        fprintf(f, "], \"material_ids\": [");
        const auto& mi = shapes[s].mesh.material_ids;
        for(size_t i = 0; i < mi.size(); ++i) fprintf(f, "%d%s", mi[i], i + 1 < mi.size() ? ", " : "");
        fprintf(f, "]}%s\n", s + 1 < shapes.size() ? "," : "");
    }
    fprintf(f, "],\n\"materials\": [\n");
    for(size_t m = 0; m < materials.size(); ++m)
    {
        const auto& t = materials[m];
        fprintf(f, "{\"name\": \"%s\", \"ambient\": [%.9g, %.9g, %.9g], \"diffuse\": [%.9g, %.9g, %.9g], "
                   "\"specular\": [%.9g, %.9g, %.9g], \"transmittance\": [%.9g, %.9g, %.9g], "
                   "\"emission\": [%.9g, %.9g, %.9g], \"shininess\": %.9g, \"ior\": %.9g, \"dissolve\": %.9g, "
                   "\"illum\": %d, \"roughness\": %.9g, \"metallic\": %.9g, \"diffuse_texname\": \"%s\", "
                   "\"specular_texname\": \"%s\", \"bump_texname\": \"%s\"}%s\n",
                esc(t.name).c_str(), t.ambient[0], t.ambient[1], t.ambient[2], t.diffuse[0], t.diffuse[1], t.diffuse[2],
                t.specular[0], t.specular[1], t.specular[2], t.transmittance[0], t.transmittance[1],
                t.transmittance[2], t.emission[0], t.emission[1], t.emission[2], t.shininess, t.ior, t.dissolve,
                t.illum, t.roughness, t.metallic,
                esc(t.diffuse_texname).c_str(), esc(t.specular_texname).c_str(), esc(t.bump_texname).c_str(), m + 1 < materials.size() ? "," : "");
    }
    fprintf(f, "], \"warn\": \"%s\", \"err\": \"%s\"}", esc(warn).c_str(), esc(err).c_str());
}

int main(int argc, char** argv)
{
    // ref_host_dump --obj <obj> <mtl dir> <out.json>: the loader's output for one more OBJ file (tests/golden/obj_cases)
    if(argc == 5 && std::string(argv[1]) == "--obj")
    {
        FILE* f = fopen(argv[4], "w");
        if(!f) return 1;
        fprintf(f, "{");
        dump_obj(f, argv[2], argv[3]);
        fprintf(f, "}\n");
        fclose(f);
        return 0;
    }
    if(argc < 4) return 2;
    FILE* f = fopen(argv[3], "w");
    if(!f) return 1;
    fprintf(f, "{\n\"camera\": [\n");
    struct Pose { float fov, aspect, znear, zfar, px, py, pz, yaw, pitch; };
    const Pose poses[] = {{65.f, 1.f, 0.01f, 100.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                          {65.f, 1920.f / 1080.f, 0.01f, 100.f, -5.2f, 1.7f, 3.3f, -32.5f, -9.0f},
                          {40.f, 4.f / 3.f, 0.1f, 1000.f, 10.f, -3.f, 7.5f, 123.f, 45.f},
                          {90.f, 0.5f, 0.5f, 50.f, 0.25f, 9.f, -2.f, 270.f, -89.f}};
    const int np = (int)(sizeof(poses) / sizeof(poses[0]));
    for(int i = 0; i < np; ++i)
    {
        const Pose& p = poses[i];
        // CameraControls::initDefaults / updateViewMatrix
        glm::mat4 projection = glm::perspective(glm::radians(p.fov), p.aspect, p.znear, p.zfar);
        glm::vec3 forward;
        forward.x = std::cos(glm::radians(p.yaw)) * std::cos(glm::radians(p.pitch));
        forward.y = std::sin(glm::radians(p.pitch));
        forward.z = std::sin(glm::radians(p.yaw)) * std::cos(glm::radians(p.pitch));
        forward   = glm::normalize(forward);
        glm::vec3 pos(p.px, p.py, p.pz);
        glm::mat4 view = glm::lookAt(pos, pos + forward, glm::vec3(0.0f, 1.0f, 0.0f));
        // vkContext::updateGraphicsUniforms
        glm::mat4 proj = projection;
        proj[1][1] *= -1.0f;
        glm::mat4 modelIT = glm::inverseTranspose(glm::mat4(1.0f));
        glm::mat4 pvi     = glm::inverse(view) * glm::inverse(proj);
        glm::mat4 light   = glm::inverse(view);
        fprintf(f, "{\"fov\": %.9g, \"aspect\": %.9g, \"near\": %.9g, \"far\": %.9g, \"pos\": [%.9g, %.9g, %.9g], "
                   "\"yaw\": %.9g, \"pitch\": %.9g,\n",
                p.fov, p.aspect, p.znear, p.zfar, p.px, p.py, p.pz, p.yaw, p.pitch);
        mat(f, "projection", projection);
        mat(f, "view", view);
        mat(f, "proj", proj);
        mat(f, "modelIT", modelIT);
        mat(f, "projViewInverse", pvi);
        mat(f, "lightTransform", light, false);
        fprintf(f, "}%s\n", i + 1 < np ? "," : "");
    }
    fprintf(f, "],\n");

    dump_obj(f, argv[1], argv[2]);
    fprintf(f, ",\n\"textures\": {\n");
    for(int i = 4; i < argc; ++i)
    {
        int      w = 0, h = 0, c = 0;
        stbi_uc* px = stbi_load(argv[i], &w, &h, &c, STBI_rgb_alpha);
        std::string name(argv[i]);
        name = name.substr(name.find_last_of('/') + 1);
        fprintf(f, "\"%s\": {\"ok\": %s, \"width\": %d, \"height\": %d, \"rgba\": [", name.c_str(), px ? "true" : "false", w, h);
        if(px)
            for(int k = 0; k < w * h * 4; ++k) fprintf(f, "%d%s", px[k], k + 1 < w * h * 4 ? "," : "");
        fprintf(f, "]}%s\n", i + 1 < argc ? "," : "");
        if(px) stbi_image_free(px);
    }
    fprintf(f, "}\n}\n");
    fclose(f);
    return 0;
}
