/* prt.h — C-ABI of the MI355X ray-tracing path (libprt_hip.so).
 *
 * Drop-in boundary: the reference's class VkRTX (/root/reference/src/vkRTX_setup.h:17-195), owned by
 * vkContext (/root/reference/src/vkContext.h:147). Every entry point below names the VkRTX member (or
 * the shader / driver call) it replaces. Plain pointers and sizes only; no exceptions cross the ABI:
 * every function returns 0 on success or a negative PRT_E_* code, and prt_last_error() gives the text
 * (the reference throws std::runtime_error from VK_CHECK_RESULT, src/vkTools.h:88-98).
 *
 * A context is bound to one GPU and is not thread-safe (the reference renders from a single thread on
 * a single VkQueue). All calls are asynchronous on the context's HIP stream unless noted.
 */
#ifndef PRT_H
#define PRT_H

#include "prt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prt_context prt_context;

#define PRT_OK 0
#define PRT_E_INVALID (-1)  /* bad argument / call order */
#define PRT_E_HIP (-2)      /* a HIP runtime call failed */
#define PRT_E_NOSCENE (-3)  /* scene or acceleration structure missing */
#define PRT_E_LIMIT (-4)    /* input exceeds a documented limit */

/* rendering modes — vkContext.h:184-185, selected in VkRTX::recordCommandBuffer (vkRTX_setup.cpp:721-764) */
#define PRT_MODE_PATH 0 /* shaders/pathRT.rgen */
#define PRT_MODE_AO 1   /* shaders/AO.rgen */

/* prt_set_flags */
#define PRT_FLAG_LITERAL 1u /* reference image semantics: E is reset when ubo.iteration <= 1 (pathRT.rgen:649-654)
                               and a primary miss OVERWRITES the pixel (pathRT.rgen:693-697). Default (0) is the
                               additive form needed for multi-GPU sums: E always accumulates, clear with
                               prt_clear_accum(). */
#define PRT_FLAG_RUSSIAN_ROULETTE 2u /* extension, not in the reference: off by default */

/* 32-byte node of the build tree, as returned by prt_read_bvh: count == 0 -> inner node with children
 * nodes[first], nodes[first+1]; count > 0 -> leaf over order[first .. first+count). */
typedef struct prt_bvh_node
{
    float    bmin[3];
    uint32_t first;
    float    bmax[3];
    uint32_t count;
} prt_bvh_node;

typedef struct prt_ray
{
    float o[3];
    float tmin;
    float d[3]; /* not normalised: t is in units of |d| (traceNV semantics) */
    float tmax;
} prt_ray;

typedef struct prt_hit
{
    uint32_t prim; /* gl_PrimitiveID, 0xffffffff on miss (pathRT.rmiss:17-20) */
    float    u, v; /* hitAttributeNV.xy (pathRT.rchit:19) */
    float    t;
} prt_hit;

typedef struct prt_stats
{
    uint64_t rays;            /* primary + bounce + shadow/AO rays traced since the last prt_reset_stats */
    uint64_t iterations;      /* prt_render_iteration equivalents executed */
    uint32_t bvh_nodes;       /* build-tree nodes */
    uint32_t bvh_inner_nodes; /* 64-byte traversal nodes */
    uint32_t bvh_depth;
    float    bvh_build_ms;    /* wall time of the last prt_build_accel */
    float    bvh_refit_ms;    /* wall time of the last prt_refit_accel */
    uint32_t trace_overruns;  /* traversal waves that hit their safety bound and dropped rays: 0 unless the
                                 acceleration structure in device memory was corrupted */
} prt_stats;

/* Kernel kinds of prt_profile */
#define PRT_K_GENERATE 0
#define PRT_K_EXTEND 1   /* closest-hit traversal (traceNV, pathRT.rgen:692,963) */
#define PRT_K_SHADE 2
#define PRT_K_SHADOW 3   /* any-hit traversal + accumulate (pathRT.rgen:818-822, 939-942) */
#define PRT_K_AO_SHADE 4 /* AO.rgen:222-272 */
#define PRT_K_RESOLVE 5  /* E += terms of the pass, in path order (pathRT.rgen:939-942, :973) */
#define PRT_K_SHADOW_EXTEND 6 /* shadow rays of bounce b and extension rays of bounce b+1 in one launch */
#define PRT_K_COUNT 8

/* Wave-level work of the traversal kernels at profiling level 2: number of wave-wide node steps / triangle tests /
 * ray set-ups executed, and the lanes that did useful work in them (lanes / (64 * steps) = SIMD utilisation). */
typedef struct prt_wave_steps
{
    uint64_t node_steps, node_lanes, tri_steps, tri_lanes, setup_steps, setup_lanes;
    /* s_memtime ticks summed over the waves: in ray set-up, node phases, triangle phases, whole kernel body */
    uint64_t setup_ticks, node_ticks, tri_ticks, total_ticks;
} prt_wave_steps;

typedef struct prt_profile
{
    double   ms[PRT_K_COUNT];       /* summed kernel time per kind, HIP events on the context's stream */
    uint64_t launches[PRT_K_COUNT];
    uint64_t closest_rays, any_rays;
    uint64_t closest_nodes, closest_tris; /* 64-byte nodes fetched / triangles tested by closest-hit rays */
    uint64_t any_nodes, any_tris;         /* ... by any-hit rays (both only at profiling level 2) */
    prt_wave_steps closest_steps, any_steps;
    uint64_t fused_closest_rays, fused_any_rays; /* the part of closest_rays / any_rays traced by PRT_K_SHADOW_EXTEND launches */
} prt_profile;

/* VkRTX::VkRTX(ctx, extent) — vkRTX_setup.h:20-24. Binds to HIP device `device_ordinal`. */
int prt_create(int device_ordinal, uint32_t width, uint32_t height, prt_context** out);
/* VkRTX::cleanUp — vkRTX_setup.h:40, vkRTX_setup.cpp:913-1032 */
int prt_destroy(prt_context* ctx);
/* Text of the last error on this context (ctx may be NULL: last error of prt_create). */
const char* prt_last_error(const prt_context* ctx);

/* VkRTX::initRaytracing(gpu, models, ubo, uboMem) — vkRTX_setup.h:26-29; uses models[0] only
 * (vkRTX_setup.cpp:67-75, 459, 581-624). Host arrays are copied; the caller keeps ownership. */
int prt_set_scene(prt_context* ctx, const prt_vertex* vertices, size_t n_vertices, const uint32_t* indices,
                  size_t n_indices, const prt_material* materials, size_t n_materials, const prt_texture* textures,
                  size_t n_textures);
/* createAccelerationStructures (vkRTX_setup.cpp:391-415) -> vkCmdBuildAccelerationStructureNV
 * (vkRT_BLAS.cpp:271, vkRT_TLAS.cpp:262): binned-SAH BVH build on the GPU. Synchronous. */
int prt_build_accel(prt_context* ctx);
/* The "updateOnly" path of the reference's acceleration-structure helpers (vkRT_BLAS.cpp:209-290 and
 * vkRT_TLAS.cpp:178-279 take `updateOnly`; vkRTX_setup.cpp:391-415 always passes false): replace the vertex
 * array (same count, same indices) and refit the boxes of the existing tree to it. Rendering between the two
 * calls is refused. Hits are the same as after a full rebuild (the trace result does not depend on the tree). */
int prt_update_vertices(prt_context* ctx, const prt_vertex* vertices, size_t n_vertices);
int prt_refit_accel(prt_context* ctx);

/* Two-level scenes: the TLAS of the reference's helper takes any number of instances with transforms
 * (TopLevelASGenerator::AddInstance, vkRT_TLAS.cpp:58-72; one bottom-level structure per geometry,
 * vkRT_BLAS.cpp:87-109); the reference itself wires one identity instance of model 0 (vkRTX_setup.cpp:138-186), which is
 * what a scene without this call is. `blas` = ranges of the scene's triangles (object space, materials as in the
 * scene), `instances` = placements of them. Call between prt_set_scene and prt_build_accel; n_instances == 0 returns to
 * the single-level scene. One BVH is built per range and shared by its instances; a top-level BVH over the instances'
 * world boxes is walked by the same kernels, a ray enters an instance in object space (inverse transform, same t).
 * gl_PrimitiveID as reported in hits / prt_read_hit_ids is flattened: instance i's triangles follow those of
 * instances 0 .. i-1 (what pathtracer_amd/host/Scene.h's flattening yields). Transforms must be invertible with a
 * condition number (row-sum norms of the 3 x 3 part) <= 1024. */
int prt_set_instances(prt_context* ctx, const prt_blas_range* blas, size_t n_blas, const prt_instance* instances,
                      size_t n_instances);
/* The updateOnly path of the top level (vkRT_TLAS.cpp:178-279 with updateOnly = true): new transforms
 * (n_instances x 16 floats, column-major) for the instances; the bottom-level structures stay. Synchronous. */
int prt_update_instance_transforms(prt_context* ctx, const float* transforms, size_t n_instances);

/* generateNewScrambles + updateScrambleValueImage — vkRTX_setup.h:38-39, vkRTX_setup.cpp:808-911.
 * Seeded, stateless per (pixel, layer) hash evaluated in the kernels (default seed 1) ... */
int prt_set_scramble_seed(prt_context* ctx, uint32_t seed);
/* ... or the reference's explicit table: n_layers (= 32) x width*height u32, layer-major. Copied. */
int prt_upload_scrambles(prt_context* ctx, const uint32_t* layers, uint32_t n_layers);

/* The host-mapped UBO that vkContext memcpy's every frame — vkContext.cpp:1057-1059. */
int prt_set_uniforms(prt_context* ctx, const prt_uniforms* ubo);
int prt_set_flags(prt_context* ctx, uint32_t flags);

/* VkRTX::recordCommandBuffer(cmd, rp, fb, img, mode) — vkRTX_setup.h:32-36: one vkCmdTraceRaysNV(W,H,1)
 * with the current uniforms == one progressive iteration for all pixels. */
int prt_render_iteration(prt_context* ctx, int mode);
/* Convenience: iterations first_iter, first_iter+stride, ... (n_iter of them); ubo.iteration is
 * overwritten for each. stride > 1 is the multi-GPU sample sharding (rank r renders first_iter = 1+r,
 * stride = world size). */
int prt_render(prt_context* ctx, int mode, uint32_t first_iter, uint32_t n_iter, uint32_t iter_stride);
/* Iterations of the path mode are independent until their radiance terms are added to the image, so up to
 * `n` of them (1..8) are in flight on separate HIP streams; the sums are still formed in iteration
 * order (bit-identical to n = 1). No counterpart in the reference (one queue, one frame at a time). */
int prt_set_pipeline_depth(prt_context* ctx, int n);
int prt_get_pipeline_depth(const prt_context* ctx); /* current depth (default: 3 when the process has >= 8 HIP
                                                       hardware queues, GPU_MAX_HW_QUEUES, else 2) */
/* prt_render additionally puts up to `n` consecutive iterations (1..64) into ONE pass over n*W*H "pixels" when
 * the frame is small (default: about 128 M pixels per pass, at most 64 iterations; passes of 24 M pixels or more run
 * one after the other on one stream, smaller ones rotate over the pipeline's streams), because a launch's fixed tail does
 * not grow with its size. Same image, bit for bit; the per-pass buffers grow n-fold. prt_render_iteration is not
 * affected. No counterpart in the reference. */
int prt_set_iteration_batch(prt_context* ctx, int n);
int prt_get_iteration_batch(const prt_context* ctx);
/* Memory policy of prt_render (no counterpart in the reference). A pass of n iterations needs about
 * (218 + 16 * bounces) bytes per pixel and iteration (38 GB at 1080p, 5 bounces, 64 iterations). When those buffers cannot
 * be allocated — hipErrorOutOfMemory, or more than `bytes` (0 = no bound; default: environment PRT_MAX_PASS_BYTES) —
 * prt_render halves the iterations per pass and tries again, down to one iteration per pass, whose buffers exist since
 * prt_create; the image is the same bit for bit. prt_get_pass_iterations: iterations per pass that the last prt_render
 * actually used (0: none rendered yet). */
int prt_set_pass_memory_limit(prt_context* ctx, uint64_t bytes);
int prt_get_pass_iterations(const prt_context* ctx);
/* vkQueueWaitIdle — vkTools.cpp:212 */
int prt_sync(prt_context* ctx);

/* m_rtRenderTarget (vkRTX_setup.h:186-194; RGBA16F there, float4 here): W*H float4, row-major,
 * pixel (x,y) at y*W+x == gl_LaunchIDNV.xy; the image pathRTpostprocess.comp binding 1 reads. */
int prt_clear_accum(prt_context* ctx);
int prt_read_accum(prt_context* ctx, float* rgba_out); /* synchronous */
/* Upload W*H float4 into the accumulation image: resume of a saved progressive render (the reference keeps the
 * image only in device memory). Synchronous. */
int prt_write_accum(prt_context* ctx, const float* rgba_in);
int prt_device_accum_ptr(prt_context* ctx, void** dev_ptr);
/* Render into caller-owned device memory (W*H*16 bytes, 16-byte aligned), e.g. a buffer that an RCCL
 * all-reduce will sum; NULL returns to the context's own buffer. */
int prt_bind_accum(prt_context* ctx, void* dev_ptr);
/* pathRTpostprocess.comp:38-51 over the full image -> RGBA8. Synchronous. */
int prt_postprocess(prt_context* ctx, uint8_t* rgba8_out);

/* Primary-ray primitive ids of the last iteration (payload.primitiveID after pathRT.rgen:692). Synchronous. */
int prt_read_hit_ids(prt_context* ctx, uint32_t* ids_out);
int prt_get_stats(prt_context* ctx, prt_stats* out); /* synchronous */
int prt_reset_stats(prt_context* ctx);
/* Measurement builds only (tools/shade_stamps.py, -DPRT_SHADE_STAMPS): the raw 48 statistics words of the context
 * (ray counts, node / triangle tallies, wave-step counts, phase time stamps). Synchronous. No counterpart in the
 * reference. */
int prt_debug_read_stats(prt_context* ctx, unsigned long long* out48);
/* The reference has no profiling hooks. level 0: off; 1: bracket every kernel launch with HIP
 * events on the context's stream; 2: additionally run the traversal kernels in their instrumented form that
 * counts node fetches and triangle tests. prt_get_profile is synchronous; totals since prt_reset_stats. */
int prt_set_profiling(prt_context* ctx, int level);
/* Restrict the event brackets to the kernel kinds in `kinds` (bit PRT_K_*; default: all). Two events per launch on
 * every kernel cost the pipelined render about a tenth of its rate; bracketing only the kernel under study does not. */
int prt_set_profiling_mask(prt_context* ctx, uint32_t kinds);
int prt_get_profile(prt_context* ctx, prt_profile* out);

/* traceNV for a batch of host rays (closest hit, or any_hit != 0 for the terminate-on-first-hit
 * occlusion query of pathRT.rgen:819-822). Synchronous; for tests and ray-level measurements. */
int prt_trace(prt_context* ctx, const prt_ray* rays, size_t n, int any_hit, prt_hit* hits_out);
/* Same on device-resident rays/hits (2 x float4 per ray, 1 x float4 per hit), timed with HIP events on
 * the context's stream; *ms_out receives the kernel time. */
int prt_trace_device(prt_context* ctx, const void* rays_dev, size_t n, int any_hit, void* hits_dev, float* ms_out);

/* prt_trace_device for callers that cannot allocate. No counterpart in the reference: it exists for callers that
 * must not call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid allocation).
 * It calls no hipMalloc / hipFree / event or stream creation; the one piece of scratch the persistent traversal
 * kernel needs (its work-fetch cursor) is a 256-byte file-scope __device__ array of the library, one copy per GPU,
 * held until the process ends. Rays/hits as in prt_trace_device (2 x float4 per ray, 1 x float4 per hit, device
 * memory of the caller). Asynchronous: the launch is enqueued on `stream` (a hipStream_t; NULL = the context's own
 * stream, then prt_sync waits for it) and not timed. n above PRT_NOALLOC_MAX_RAYS is refused with PRT_E_LIMIT.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams or contexts
 * without ordering between them) race on the cursor; serialise them.
 * The launch carries the acceleration structure by value (array addresses, root, quantisation grid): a launch that was
 * captured into a graph, or is still in flight on the caller's stream, is INVALIDATED by prt_build_accel,
 * prt_refit_accel, prt_update_instance_transforms, prt_set_scene, prt_set_instances and prt_destroy. The library cannot
 * see the caller's stream: order it before those calls (hipStreamSynchronize), and capture again after them. (A refit or
 * an instance update that does not grow the structure keeps the arrays where they are, so a stale launch reads valid
 * memory — but with the old grid it decodes the new boxes wrongly: its hits are not to be used.) */
#define PRT_NOALLOC_MAX_RAYS 0x40000000u /* 2^30 rays per call */
int prt_trace_device_noalloc(prt_context* ctx, const void* rays_dev, size_t n, int any_hit, void* hits_dev,
                             void* stream);

/* Inspection of the acceleration structure (nodes_out: bvh_nodes entries, order_out: one per triangle). */
int prt_read_bvh(prt_context* ctx, prt_bvh_node* nodes_out, uint32_t* order_out);
/* The same for bottom-level structure `blas` of a two-level scene (order: ids local to its range). nodes_out /
 * order_out may be NULL; *n_nodes_out (optional) receives the node count. */
int prt_read_blas_bvh(prt_context* ctx, uint32_t blas, prt_bvh_node* nodes_out, uint32_t* order_out,
                      uint32_t* n_nodes_out);
/* The 64 x 32 Sobol matrix words in use (sobol.h:28-33 restricted to what the path reads). */
int prt_read_sobol(prt_context* ctx, uint32_t* words_out);

/* ---- several GPUs in one process (SURVEY.md 8b "Multi-GPU", 8e) -------------------------------------------
 * No counterpart in the reference (one VkDevice, one queue). One prt_context per listed HIP device, one host
 * thread per member while rendering; iterations are dealt out round-robin (member r of G renders first+r,
 * first+r+G, ...: the sample sharding of SURVEY 8e) and ONE exchange step sums the members' images into the
 * root's: RCCL ncclReduce over xGMI when the devices are distinct (librccl.so is loaded at that moment), peer
 * copies + an add kernel when a device is listed twice or PRT_GROUP_NO_RCCL=1. After prt_group_reduce the root
 * holds the total and the other members' images are cleared, so render / reduce can be repeated.
 * Across G the image differs only by fp32 summation order. Use the default additive miss semantics. */
typedef struct prt_group prt_group;
int prt_group_create(int n, const int* device_ordinals, uint32_t width, uint32_t height, prt_group** out);
int prt_group_destroy(prt_group* g);
const char* prt_group_last_error(const prt_group* g); /* g may be NULL: last error of prt_group_create */
int prt_group_size(const prt_group* g);
int prt_group_uses_rccl(const prt_group* g);
prt_context* prt_group_context(prt_group* g, int member); /* for per-member calls (profiling, inspection) */
/* prt_set_scene + prt_build_accel on every member (replicated scene and BVH) */
int prt_group_set_scene(prt_group* g, const prt_vertex* vertices, size_t n_vertices, const uint32_t* indices,
                        size_t n_indices, const prt_material* materials, size_t n_materials,
                        const prt_texture* textures, size_t n_textures);
/* prt_set_instances + prt_build_accel on every member (after prt_group_set_scene) */
int prt_group_set_instances(prt_group* g, const prt_blas_range* blas, size_t n_blas, const prt_instance* instances,
                            size_t n_instances);
int prt_group_set_scramble_seed(prt_group* g, uint32_t seed);
int prt_group_set_uniforms(prt_group* g, const prt_uniforms* ubo);
int prt_group_set_flags(prt_group* g, uint32_t flags);
int prt_group_clear_accum(prt_group* g);
/* iterations first_iter .. first_iter+n_iter-1, sharded over the members; returns when all have finished */
int prt_group_render(prt_group* g, int mode, uint32_t first_iter, uint32_t n_iter);
int prt_group_reduce(prt_group* g, int root);
int prt_group_read_accum(prt_group* g, int root, float* rgba_out);
int prt_group_get_stats(prt_group* g, prt_stats* out); /* rays and iterations summed over the members */

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */
