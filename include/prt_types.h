/* prt_types.h — plain-C data contracts of the ray-tracing hot path.
 *
 * These three structs are the byte-level interface between the reference's host code and its
 * device programs; the MI355X path keeps them bit-for-bit so that buffers prepared by the
 * reference's `Model` / `vkContext` can be handed over unchanged.
 *
 *   prt_vertex    <- VkTools::VertexPNTC            /root/reference/src/Model.h:53-59
 *                    (GLSL view: 3 x vec4, shaders/pathRT.rgen:94-119)
 *   prt_material  <- VkTools::Material              /root/reference/src/Model.h:25-41
 *                    (GLSL view: 6 x vec4, shaders/pathRT.rgen:125-169)
 *   prt_uniforms  <- vkContext::UniformBufferObject /root/reference/src/vkContext.h:54-82
 *                    (GLSL view: std140 UBO, shaders/pathRT.rgen:13-43)
 */
#ifndef PRT_TYPES_H
#define PRT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 48 bytes. p, n, t, c, materialID — Model.h:53-59. */
typedef struct prt_vertex
{
    float   p[3];
    float   n[3];
    float   t[2];
    float   c[3];
    int32_t materialID;
} prt_vertex;

/* 96 bytes — Model.h:25-41. `illum` and the three texture ids are int bit patterns that the
 * shader re-reads with floatBitsToInt (pathRT.rgen:164-167). */
typedef struct prt_material
{
    float   ambient[3];
    float   diffuse[3];
    float   specular[3];
    float   transmittance[3];
    float   emission[3];
    float   shininess;
    float   metallic;
    float   ior;
    float   dissolve;
    int32_t illum;
    int32_t diffuseTextureID;
    int32_t specularTextureID;
    int32_t normalTextureID;
    float   pad;
} prt_material;

/* 464 bytes, column-major mat4 (glm / GLSL convention: m[col][row]) — vkContext.h:54-82. */
typedef struct prt_uniforms
{
    float    model[16];           /*   0 */
    float    view[16];            /*  64 */
    float    proj[16];            /* 128 */
    float    modelIT[16];         /* 192 */
    float    projViewInverse[16]; /* 256  (GLSL name: viewProjInverse) */
    float    lightTransform[16];  /* 320 */
    float    lightSize[2];        /* 384 */
    float    pad0[2];             /* 392 */
    float    lightE[3];           /* 400 */
    float    pad1;                /* 412 */
    int32_t  numIndirectBounces;  /* 416 */
    int32_t  samplesPerPixel;     /* 420 */
    float    lightSourceArea;     /* 424 */
    float    lightOtherE;         /* 428 */
    int32_t  numAArays;           /* 432 */
    float    filterRadius;        /* 436 */
    int32_t  numAOrays;           /* 440 */
    float    aoRayLength;         /* 444 */
    uint32_t iteration;           /* 448 */
    float    time;                /* 452 */
    float    pad2[2];             /* 456  (struct is 16-byte aligned through glm::mat4) */
} prt_uniforms;

/* RGBA8 UNORM texture, row-major, as stb_image hands it to Model::createTextures
 * (/root/reference/src/Model.cpp:311-387). Sampled bilinear / REPEAT / no mips
 * (/root/reference/src/vkTools.cpp:405,456-480). */
typedef struct prt_texture
{
    const uint8_t* rgba;
    uint32_t       width;
    uint32_t       height;
} prt_texture;

/* Two-level scenes. A bottom-level structure is a range of the scene's triangles (object space); an instance places
 * one of them in the world — TopLevelASGenerator::AddInstance(blas, transform, instanceID, hitGroupIndex),
 * /root/reference/external/NVIDIA_RTX/vkRT_TLAS.cpp:58-72. `transform` is a column-major mat4 (glm) object -> world; as
 * in VkGeometryInstance (vkRT_TLAS.cpp:212-214) only its 3 x 4 affine part exists: the fourth row must be (0,0,0,1).
 * instance_id (gl_InstanceCustomIndexNV) and hit_group (SBT offset) are carried; the path's shaders read neither. */
typedef struct prt_blas_range
{
    uint32_t first_triangle;
    uint32_t n_triangles;
} prt_blas_range;

typedef struct prt_instance
{
    float    transform[16];
    uint32_t blas;        /* index into the prt_blas_range array */
    uint32_t instance_id;
    uint32_t hit_group;
    uint32_t pad;
} prt_instance;

#ifdef __cplusplus
static_assert(sizeof(prt_instance) == 80, "prt_instance is 80 bytes");
static_assert(sizeof(prt_vertex) == 48, "VertexPNTC is 48 bytes");
static_assert(sizeof(prt_material) == 96, "Material is 96 bytes");
static_assert(sizeof(prt_uniforms) == 464, "UniformBufferObject is 464 bytes");
static_assert(offsetof(prt_uniforms, lightSize) == 384, "ubo.lightSize");
static_assert(offsetof(prt_uniforms, lightE) == 400, "ubo.lightE");
static_assert(offsetof(prt_uniforms, numIndirectBounces) == 416, "ubo.numIndirectBounces");
static_assert(offsetof(prt_uniforms, numAArays) == 432, "ubo.numAArays");
static_assert(offsetof(prt_uniforms, iteration) == 448, "ubo.iteration");
static_assert(offsetof(prt_uniforms, time) == 452, "ubo.time");
}
#endif

#endif /* PRT_TYPES_H */
