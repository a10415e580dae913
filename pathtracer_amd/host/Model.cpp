// Model.cpp — VkTools::Model::LoadModelFromFile / createTextures (/root/reference/src/Model.cpp:47-221,311-387).
#include "Model.h"

#include <cstring>

#include "obj_loader.h"
#include "prt_math.h"

namespace prt_host {

Material defaultMaterial()
{
    Material m;
    memset(&m, 0, sizeof m);
    m.ambient[0] = m.ambient[1] = m.ambient[2] = 0.1f;
    m.diffuse[0] = 0.0f; m.diffuse[1] = 1.0f; m.diffuse[2] = 1.0f;
    m.specular[0] = m.specular[1] = m.specular[2] = 1.0f;
    m.shininess = 0.0f;
    m.metallic  = 0.0f;
    m.ior       = 1.0f;
    m.dissolve  = 1.0f;
    m.illum     = 0;
    m.diffuseTextureID = m.specularTextureID = m.normalTextureID = -1;
    m.pad = 0.0f;  // uninitialised in the reference; never read by the shaders
    return m;
}

static std::string replaceSubString(const std::string& s, const std::string& from, const std::string& to)
{
    std::string out = s;
    size_t      pos = 0;
    while((pos = out.find(from, pos)) != std::string::npos)
    {
        out.replace(pos, from.length(), to);
        pos += to.length();
    }
    return out;
}

// Model.cpp:93-148 — a texture name is looked up (as written in the MTL) among the paths stored so far (after
// the "\\\\" -> "/" replacement), appended when new
static int textureSlot(std::vector<std::string>& paths, const std::string& name, int fallback)
{
    for(size_t i = 0; i < paths.size(); ++i)
        if(std::strcmp(paths[i].c_str(), name.c_str()) == 0) return (int)i;
    if(!name.empty())
    {
        paths.push_back(replaceSubString(name, "\\\\", "/"));
        return (int)(paths.size() - 1);
    }
    return fallback;
}

void Model::LoadModelFromFile(const std::string& filepath)
{
    ObjData     obj;
    std::string err;
    if(!load_obj(filepath, directory, obj, err)) throw std::runtime_error(obj.warn + err);
    warnings = obj.warn;

    numVertices = numIndices = 0;
    for(const ObjShape& s : obj.shapes)
    {
        numVertices += s.indices.size();
        numIndices += s.indices.size();
    }
    m_vertices.reserve(numVertices);
    m_indices.reserve(numIndices);

    for(const ObjMaterial& mat : obj.materials)
    {
        Material m = defaultMaterial();
        for(int k = 0; k < 3; ++k)
        {
            m.ambient[k]       = mat.ambient[k];
            m.diffuse[k]       = mat.diffuse[k];
            m.specular[k]      = mat.specular[k];
            m.emission[k]      = mat.emission[k];
            m.transmittance[k] = mat.transmittance[k];
        }
        m.dissolve = mat.dissolve;
        m.ior      = mat.ior;
        m.illum    = mat.illum;
        // shininess comes from the PBR roughness `Pr`, not from `Ns` (Model.cpp:83-90)
        m.shininess = mat.roughness == 0.0f ? 0.0f : (float)(1.0 - mat.roughness);
        m.diffuseTextureID  = textureSlot(m_texturePaths, mat.diffuse_texname, m.diffuseTextureID);
        m.specularTextureID = textureSlot(m_texturePaths, mat.specular_texname, m.specularTextureID);
        m.normalTextureID   = textureSlot(m_texturePaths, mat.bump_texname, m.normalTextureID);
        m_materials.push_back(m);
    }
    if(m_materials.empty()) m_materials.push_back(defaultMaterial());

    for(const ObjShape& shape : obj.shapes)
    {
        uint32_t faceID     = 0;
        int      indexCount = 0;
        for(const ObjIndex& index : shape.indices)
        {
            VertexPNTC vertex;
            memset(&vertex, 0, sizeof vertex);
            vertex.materialID = -1;
            const float* vpos = &obj.vertices[3 * (size_t)index.vertex_index];
            vertex.p[0] = vpos[0]; vertex.p[1] = vpos[1]; vertex.p[2] = vpos[2];
            if(!obj.normals.empty() && index.normal_index >= 0)
            {
                const float* n = &obj.normals[3 * (size_t)index.normal_index];
                vertex.n[0] = n[0]; vertex.n[1] = n[1]; vertex.n[2] = n[2];
            }
            if(!obj.texcoords.empty() && index.texcoord_index >= 0)
            {
                const float* t = &obj.texcoords[2 * (size_t)index.texcoord_index];
                vertex.t[0] = t[0];
                vertex.t[1] = -t[1];  // Model.cpp:180
            }
            if(!obj.colors.empty())
            {
                const float* c = &obj.colors[3 * (size_t)index.vertex_index];
                vertex.c[0] = c[0]; vertex.c[1] = c[1]; vertex.c[2] = c[2];
            }
            vertex.materialID = shape.material_ids[faceID];
            if(vertex.materialID < 0 || (size_t)vertex.materialID >= m_materials.size()) vertex.materialID = 0;
            indexCount++;
            if(indexCount >= 3)
            {
                faceID++;
                indexCount = 0;
            }
            m_vertices.push_back(vertex);
            m_indices.push_back((uint32_t)m_indices.size());
        }
    }

    if(obj.normals.empty())  // flat normals only when the file has no normals at all (Model.cpp:207-220)
        for(size_t i = 0; i + 2 < m_indices.size(); i += 3)
        {
            VertexPNTC &v0 = m_vertices[m_indices[i]], &v1 = m_vertices[m_indices[i + 1]], &v2 = m_vertices[m_indices[i + 2]];
            const vec3 p0(v0.p[0], v0.p[1], v0.p[2]), p1(v1.p[0], v1.p[1], v1.p[2]), p2(v2.p[0], v2.p[1], v2.p[2]);
            const vec3 n = normalize(cross(p1 - p0, p2 - p0));
            for(VertexPNTC* v : {&v0, &v1, &v2})
            {
                v->n[0] = n.x; v->n[1] = n.y; v->n[2] = n.z;
            }
        }
}

void Model::createTextures()
{
    auto white = [](Texture& t) {
        t.width = t.height = 1;
        t.image.width = t.image.height = 1;
        t.image.rgba.assign(4, 255);
    };
    if(m_texturePaths.empty())  // Model.cpp:313-343: one 1x1 white texture so that binding 7 is never empty
    {
        Texture t;
        white(t);
        m_textures.push_back(t);
        return;
    }
    for(const std::string& texture : m_texturePaths)
    {
        Texture t;
        t.id   = (uint32_t)m_textures.size();
        t.path = texture;
        if(load_image(directory + '/' + texture, t.image))
        {
            t.width  = t.image.width;
            t.height = t.image.height;
        }
        else
        {
            white(t);  // Model.cpp:352-359
            warnings += "texture [" + directory + '/' + texture + "] could not be decoded: 1x1 white used\n";
        }
        m_textures.push_back(t);
    }
}

}  // namespace prt_host
