// Scene.h — several models / several instances of them in one ray-traced scene.
// The reference wires exactly one instance of models[0] with the identity transform into its top-level structure
// (/root/reference/src/vkRTX_setup.cpp:138-186: one Instance{blas 0, instanceId 0, hitGroup 0, mat4(1)}), but the
// NVIDIA helper it calls takes any number of them (external/NVIDIA_RTX/vkRT_TLAS.cpp:58-72 AddInstance(blas,
// transform, instanceID, hitGroupIndex)). Two ways to the device:
//  * poolInstances: the models side by side in ONE object-space geometry pool + one bottom-level range per model +
//    the instance list — what prt_set_scene + prt_set_instances take (two-level traversal on the GPU: one BVH per
//    model, shared by its instances). The default of Renderer::initRaytracing(models, instances).
//  * flattenInstances: the instances copied into ONE world-space Model (what SURVEY.md 8f rank 4 and config 5,
//    "instanced-then-flattened", describe); single-level device path, memory grows with the instance count.
// gl_PrimitiveID is the flattened triangle index in both. pathRT.rgen never reads gl_InstanceID /
// gl_InstanceCustomIndexNV, so nothing else changes.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/prt_types.h"
#include "Model.h"
#include "prt_math.h"

namespace prt_host {

struct Instance  // vkRT_TLAS.h Instance / VkGeometryInstanceNV
{
    uint32_t modelIndex    = 0;           // which bottom-level structure (= which Model)
    mat4     transform     = mat4(1.0f);  // object -> world (column-major, glm)
    uint32_t instanceId    = 0;           // gl_InstanceCustomIndexNV (carried, unused by the shaders)
    uint32_t hitGroupIndex = 0;           // SBT offset (carried; the path has one hit group)
};

struct FlatScene
{
    Model                 model;          // world-space vertices, identity indices, merged materials / textures
    std::vector<uint32_t> firstTriangle;  // per instance: first flattened triangle (gl_PrimitiveID base);
                                          // one extra entry = total
    std::vector<uint32_t> instanceId;     // per instance, as given
};

struct PooledScene
{
    Model                       model;      // object-space vertices of all models, merged materials / textures
    std::vector<prt_blas_range> ranges;     // per model: its triangles in the pool
    std::vector<prt_instance>   instances;  // as given (blas = modelIndex)
};
// Material / texture ids rebased exactly as flattenInstances does. Throws like flattenInstances.
PooledScene poolInstances(const std::vector<Model>& models, const std::vector<Instance>& instances);

// Positions: (M * vec4(p, 1)).xyz with glm's operation order; normals: normalize(mat3(transpose(inverse(M))) * n)
// (zero normals stay zero); texture coordinates and colours unchanged; materialID and the three texture ids of
// every material are rebased onto the merged tables. An instance whose transform is exactly the identity is
// copied bit for bit. Throws std::runtime_error on an out-of-range modelIndex or an empty instance list.
FlatScene flattenInstances(const std::vector<Model>& models, const std::vector<Instance>& instances);

}  // namespace prt_host
