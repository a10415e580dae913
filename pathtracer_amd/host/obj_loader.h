// obj_loader.h — Wavefront OBJ / MTL reader with the semantics of the loader the reference links
// (tinyobjloader 2.0.0-era single header, external/tinyobjloader/tiny_obj_loader.h, called at
// /root/reference/src/Model.cpp:54 with triangulation on and the default-colour fallback on).
// Own implementation; behaviour pinned by tests/golden/host_golden.json (output of that loader).
#pragma once
#include <string>
#include <vector>

namespace prt_host {

struct ObjIndex
{
    int vertex_index, normal_index, texcoord_index;  // -1: absent
};
struct ObjShape
{
    std::string           name;
    std::vector<ObjIndex> indices;       // 3 per triangle
    std::vector<int>      material_ids;  // 1 per triangle, -1: none
};
struct ObjMaterial
{
    std::string name;
    float       ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, transmittance[3] = {0, 0, 0},
          emission[3] = {0, 0, 0};
    float       shininess = 1.0f, ior = 1.0f, dissolve = 1.0f;
    int         illum = 0;
    float       roughness = 0.0f, metallic = 0.0f;
    std::string ambient_texname, diffuse_texname, specular_texname, specular_highlight_texname, bump_texname,
        displacement_texname, alpha_texname, reflection_texname, normal_texname;
};
struct ObjData
{
    std::vector<float>       vertices, normals, texcoords, colors;  // 3, 3, 2, 3 floats per element
    std::vector<ObjShape>    shapes;
    std::vector<ObjMaterial> materials;
    std::string              warn;
};

// Returns false (and sets err) when the OBJ file cannot be opened or a face has an index of 0.
bool load_obj(const std::string& filename, const std::string& mtl_basedir, ObjData& out, std::string& err);

}  // namespace prt_host
