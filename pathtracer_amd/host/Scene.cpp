// Scene.cpp — see Scene.h.
#include "Scene.h"

#include <cstring>

namespace prt_host {

static bool isIdentity(const mat4& m)
{
    const mat4 id(1.0f);
    return memcmp(m.data(), id.data(), 16 * sizeof(float)) == 0;
}

// detail/type_mat4x4.inl operator*(mat4, vec4): (m0*x + m1*y) + (m2*z + m3*w)
static vec4 mulPoint(const mat4& m, const vec4& v)
{
    const vec4 add0 = m[0] * v.x + m[1] * v.y;
    const vec4 add1 = m[2] * v.z + m[3] * v.w;
    return add0 + add1;
}
// detail/type_mat3x3.inl operator*(mat3, vec3) on the upper-left 3x3 of a mat4
static vec3 mulDir(const mat4& m, const vec3& v)
{
    return vec3(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
}

// merged material / texture tables, in model order
static void mergeTables(const std::vector<Model>& models, Model& dst, std::vector<int32_t>& matBase)
{
    std::vector<int32_t> texBase(models.size());
    matBase.resize(models.size());
    for(size_t k = 0; k < models.size(); ++k)
    {
        const Model& m = models[k];
        matBase[k]     = (int32_t)dst.m_materials.size();
        texBase[k]     = (int32_t)dst.m_textures.size();
        for(Material mat : m.m_materials)
        {
            if(mat.diffuseTextureID >= 0) mat.diffuseTextureID += texBase[k];
            if(mat.specularTextureID >= 0) mat.specularTextureID += texBase[k];
            if(mat.normalTextureID >= 0) mat.normalTextureID += texBase[k];
            dst.m_materials.push_back(mat);
        }
        for(Texture t : m.m_textures)
        {
            t.id += (uint32_t)texBase[k];
            dst.m_textures.push_back(t);
        }
        for(const std::string& p : m.m_texturePaths) dst.m_texturePaths.push_back(p);
        if(!m.warnings.empty()) dst.warnings += m.warnings;
    }
    dst.directory = models[0].directory;
}

PooledScene poolInstances(const std::vector<Model>& models, const std::vector<Instance>& instances)
{
    if(models.empty()) throw std::runtime_error("poolInstances: no model");
    if(instances.empty()) throw std::runtime_error("poolInstances: no instance");
    PooledScene          out;
    Model&               dst = out.model;
    std::vector<int32_t> matBase;
    mergeTables(models, dst, matBase);
    for(size_t k = 0; k < models.size(); ++k)
    {
        const Model&   m    = models[k];
        const uint32_t base = (uint32_t)dst.m_vertices.size();
        out.ranges.push_back(prt_blas_range{(uint32_t)(dst.m_indices.size() / 3), (uint32_t)(m.m_indices.size() / 3)});
        for(VertexPNTC v : m.m_vertices)
        {
            v.materialID += matBase[k];
            dst.m_vertices.push_back(v);
        }
        for(uint32_t i : m.m_indices) dst.m_indices.push_back(base + i);
    }
    for(const Instance& inst : instances)
    {
        if(inst.modelIndex >= models.size()) throw std::runtime_error("poolInstances: modelIndex out of range");
        prt_instance pi{};
        memcpy(pi.transform, inst.transform.data(), 64);
        pi.blas        = inst.modelIndex;
        pi.instance_id = inst.instanceId;
        pi.hit_group   = inst.hitGroupIndex;
        out.instances.push_back(pi);
    }
    dst.numVertices = dst.m_vertices.size();
    dst.numIndices  = dst.m_indices.size();
    return out;
}

FlatScene flattenInstances(const std::vector<Model>& models, const std::vector<Instance>& instances)
{
    if(models.empty()) throw std::runtime_error("flattenInstances: no model");
    if(instances.empty()) throw std::runtime_error("flattenInstances: no instance");
    FlatScene            out;
    Model&               dst = out.model;
    std::vector<int32_t> matBase;
    mergeTables(models, dst, matBase);
    for(const Instance& inst : instances)
    {
        if(inst.modelIndex >= models.size()) throw std::runtime_error("flattenInstances: modelIndex out of range");
        const Model&   m    = models[inst.modelIndex];
        const uint32_t base = (uint32_t)dst.m_vertices.size();
        out.firstTriangle.push_back((uint32_t)(dst.m_indices.size() / 3));
        out.instanceId.push_back(inst.instanceId);
        const bool ident = isIdentity(inst.transform);
        const mat4 nrm   = ident ? mat4(1.0f) : inverseTranspose(inst.transform);
        for(VertexPNTC v : m.m_vertices)
        {
            if(!ident)
            {
                const vec4 p = mulPoint(inst.transform, vec4(v.p[0], v.p[1], v.p[2], 1.0f));
                v.p[0] = p.x; v.p[1] = p.y; v.p[2] = p.z;
                vec3 n = mulDir(nrm, vec3(v.n[0], v.n[1], v.n[2]));
                if(dot(n, n) > 0.0f) n = normalize(n);
                v.n[0] = n.x; v.n[1] = n.y; v.n[2] = n.z;
            }
            v.materialID += matBase[inst.modelIndex];
            dst.m_vertices.push_back(v);
        }
        for(uint32_t i : m.m_indices) dst.m_indices.push_back(base + i);
    }
    out.firstTriangle.push_back((uint32_t)(dst.m_indices.size() / 3));
    dst.numVertices = dst.m_vertices.size();
    dst.numIndices  = dst.m_indices.size();
    return out;
}

}  // namespace prt_host
