/* oracle.h — TEST INFRASTRUCTURE, not product code.
 *
 * C interface of the scalar, multithreaded CPU restatement of the reference's ray-tracing hot path
 * (shaders/pathRT.rgen + pathRT.rchit/.rmiss + pathRTBounce.rmiss, shaders/AO.rgen + stubs,
 * sobol.h, and the driver-side BVH build / traceNV that VK_NV_ray_tracing hides).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (pathtracer_amd/, libprt_hip.so) never includes, links or calls it.
 *
 * PARITY STATUS: the Sobol sampler is pinned by the reference's own table and sampler
 * (tests/golden/sobol_*, generated from /root/reference/external/NVIDIA_RTX/sobol). Everything
 * that the reference delegates to the closed NVIDIA driver (BVH build, traversal, triangle test,
 * texture filtering) and the GLSL built-ins with implementation-defined rounding is
 * "parity unpinned": the reference holds no golden vectors, tests or CPU path for them
 * (SURVEY.md §4, §8c), so this oracle's behaviour is the definition.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/prt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* 32-byte build node, shared comparison format with the product's prt_read_bvh():
 * count == 0 -> inner node, children are nodes[first] and nodes[first+1];
 * count  > 0 -> leaf over order[first .. first+count). */
typedef struct orc_node
{
    float    bmin[3];
    uint32_t first;
    float    bmax[3];
    uint32_t count;
} orc_node;

typedef struct orc_hit
{
    uint32_t prim; /* 0xffffffff on miss (pathRT.rmiss:17-20) */
    float    u, v; /* hitAttributeNV .x/.y: barycentric weights of vertex 1 and vertex 2 */
    float    t;
} orc_hit;

typedef struct orc_ray
{
    float o[3];
    float tmin;
    float d[3];
    float tmax;
} orc_ray;

/* flags for orc_render */
#define ORC_FLAG_LITERAL 1u       /* reference semantics: E reset when iteration<=1, primary miss overwrites */
#define ORC_FLAG_RUSSIAN_ROULETTE 2u
#define ORC_FLAG_BRUTE_FORCE 4u   /* trace by testing every triangle (no BVH) */

orc_scene* orc_scene_create(const prt_vertex* v, size_t nv, const uint32_t* idx, size_t nidx,
                            const prt_material* m, size_t nm, const prt_texture* tex, size_t ntex);
void       orc_scene_destroy(orc_scene*);

/* Sobol matrices: `dims` x `size` u32 words, row-major (sobol.h:28-33). Copied. */
void orc_set_sobol(orc_scene*, const uint32_t* matrices, uint32_t dims, uint32_t size);
/* Scrambles: either a seed for the stateless hash, or an explicit 32 x (w*h) table
 * (vkRTX_setup.cpp:808-832 layout: layer-major). Table pointer is borrowed. */
void orc_set_scramble_seed(orc_scene*, uint32_t seed);
void orc_set_scramble_table(orc_scene*, const uint32_t* table, uint32_t w, uint32_t h);

/* Two-level scenes (vkRT_TLAS.cpp:58-72 AddInstance; the reference itself wires one identity instance,
 * vkRTX_setup.cpp:138-186): `blas` = ranges of the scene's triangles in object space, `inst` = instances of them.
 * A ray is taken to object space with the inverse transform (same t) and tested there; the scene's hit is the smallest
 * t over all instances, ties -> lower flattened primitive id (instance i's triangles follow those of instances
 * 0..i-1). n_inst == 0 returns to the single-level scene. Call orc_build_bvh afterwards. 0, or -1 on a bad range. */
int  orc_set_instances(orc_scene*, const prt_blas_range* blas, size_t n_blas, const prt_instance* inst, size_t n_inst);
void orc_inverse4(const float* m16, float* out16); /* glm::inverse(mat4), column-major */
uint32_t orc_blas_num_nodes(const orc_scene*, uint32_t k);
void     orc_blas_read(const orc_scene*, uint32_t k, orc_node* nodes_out, uint32_t* order_out);

int      orc_build_bvh(orc_scene*);
uint32_t orc_bvh_num_nodes(const orc_scene*);
void     orc_bvh_read(const orc_scene*, orc_node* nodes_out, uint32_t* order_out);
uint32_t orc_bvh_depth(const orc_scene*);

/* unit-level entry points used by the parity tests */
float    orc_sobol_sample(const orc_scene*, uint32_t index, uint32_t dim, uint32_t scramble);
uint32_t orc_scramble(const orc_scene*, uint32_t px, uint32_t py, uint32_t w, uint32_t h, uint32_t layer);
void     orc_sincos(float x, float* s, float* c);
void     orc_primary_ray(const prt_uniforms* ubo, uint32_t px, uint32_t py, uint32_t w, uint32_t h,
                         float varx, float vary, orc_ray* out);
void     orc_trace_closest(const orc_scene*, const orc_ray* rays, size_t n, orc_hit* out, uint32_t flags,
                           int threads);
void     orc_trace_any(const orc_scene*, const orc_ray* rays, size_t n, uint8_t* occluded, uint32_t flags,
                       int threads);

/* One call == one vkCmdTraceRaysNV(w,h,1) of the given mode (0 = pathRT, 1 = AO) with `ubo`
 * (vkRTX_setup.cpp:721-764). `accum` is the W*H float4 image (row-major), read and written.
 * `hit_ids`, if non-NULL, receives the primary-ray primitive id per pixel (last aaRay).
 * Returns the number of rays traced (primary + shadow/AO + bounce). */
uint64_t orc_render(const orc_scene*, const prt_uniforms* ubo, uint32_t w, uint32_t h, int mode,
                    uint32_t flags, float* accum, uint32_t* hit_ids, int threads);

/* pathRTpostprocess.comp:38-51 over the full image. out = RGBA8. */
void orc_postprocess(const float* accum, uint32_t w, uint32_t h, uint8_t* rgba8_out);
/* texture(textureSamplers[id], uv).xyz: bilinear, REPEAT, RGBA8 UNORM (vkTools.cpp:405,456-480); white for a bad id */
void orc_texture_rgb(const orc_scene*, int id, float u, float v, float* rgb_out);
float orc_gamma(float x);
/* the shading helpers of pathRT.rgen one at a time (identity tests, transcription audit); returns the number of floats
 * written to `out`, -1 for an unknown `which` */
enum
{
    ORC_H_FORM_BASIS = 0,          /* :176-197  in n[3]                                   out mat3 (columns T, B, n) */
    ORC_H_SAMPLE_LIGHT,            /* :304-309  in lightSize[2], lightTransform[16], prng[2]   out pdf, p[3] */
    ORC_H_GGX_DISTRIBUTION,        /* :331-341  in wm[3], alpha */
    ORC_H_GGX_G1,                  /* :343-356  in wv[3], wm[3], alpha */
    ORC_H_FRESNEL_DIELECTRIC,      /* :435-463  in cosThetaI, ni, nt */
    ORC_H_SCHLICK_FRESNEL,         /* :465-469  in r0[3], radians                         out vec3 */
    ORC_H_SMITH_MASKING,           /* :471-478  in wi[3], wo[3], alpha */
    ORC_H_SMITH_MASKING_SHADOWING, /* :480-490  in wi[3], wo[3], alpha */
    ORC_H_SAMPLE_VNDF,             /* :492-509  in wo[3], alpha, rnd[2]                   out vec3 */
    ORC_H_REFRACT,                 /* :511-526  in wi[3], wn[3], eta                      out ok, wt[3] */
    ORC_H_PRIMARY_RAY,             /* :541-567  in projViewInverse[16], px, py, w, h, var[2]   out origin[3], dir[3] */
    ORC_H_MITCHELL,                /* :575-593  in v */
    ORC_H_MITCHELL_WEIGHT,         /* :595-598  in offset[2] */
    ORC_H_COUNT
};
int orc_helper(int which, const float* in, float* out); /* the pow(x, 1/2.2) of the post-process, as defined in oracle.cpp */

#ifdef __cplusplus
}
#endif
#endif
