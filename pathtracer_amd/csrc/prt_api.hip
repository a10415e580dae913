// prt_api.hip — libprt_hip.so: the C-ABI of include/prt.h over the HIP kernels (gfx950).
// Host side of what VkRTX does in the reference (src/vkRTX_setup.{h,cpp}): owns the device copies of
// the scene, the acceleration structure, the Sobol table, the per-path state and the accumulation
// image, and issues one fixed sequence of kernel launches per iteration on one HIP stream.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/prt.h"
#include "../host/prt_math.h"  // glm's inverse(mat4) with the same operation order (pinned against glm: tests/golden)
#include "prt_bvh.hpp"
#include "prt_kernels.hpp"
#include "sobol_table.hpp"

using namespace prt;

static std::string g_createError;

// ---- scratch of the allocation-free entry point (prt_trace_device_noalloc) ------------------------------------
// The persistent traversal kernel needs one piece of device memory besides its inputs and outputs: the work-fetch
// cursor its waves pull ray chunks from (prt_trace_device takes it from hipMalloc'ed context memory). Here it is a
// file-scope __device__ array, so a caller that must not allocate (HIP graph capture, serving loops) needs no
// workspace. 256 bytes per GPU, held until the process ends. In host code `scratch` is only a stand-in for the
// symbol: its device address is asked for once per device (scratch_for_device) and only that pointer is used.
#define PRT_NOALLOC_SCRATCH_BYTES 256
static __device__ __align__(256) uint32_t scratch[PRT_NOALLOC_SCRATCH_BYTES / 4];
static constexpr int kMaxDevices = 64;
static uint32_t*     g_scratchPtr[kMaxDevices] = {};  // device address of `scratch` on every GPU the library has used
static std::mutex    g_scratchMutex;

// with `device` current: the address of this GPU's copy of `scratch` (each GPU holds its own copy of the array)
static hipError_t scratch_for_device(int device)
{
    if(device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if(g_scratchPtr[device]) return hipSuccess;
    uint32_t*        ptr = nullptr;
    const hipError_t e   = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if(e != hipSuccess) return e;
    if(!ptr) return hipErrorInvalidSymbol;
    g_scratchPtr[device] = ptr;
    return hipSuccess;
}

static constexpr uint32_t kStatWords    = 48;  // [0..28]: prt_kernels.hpp FrameView::stats; [36..42]: PRT_SHADE_STAMPS builds
static constexpr uint32_t kCounterWords = 256;
// UI cap is 10 (vkContext.cpp:266). The shader indexes scrambleArray[16] with 1 + bounce (pathRT.rgen:635-642, :703) and
// takes Sobol dimensions 4 + 4*bounce .. +3 of the 64 kept here: both are defined up to 14 bounces.
static constexpr uint32_t kMaxBounces   = 14;
static constexpr int      kMaxPipes     = 8;

// Per-pass state: one iteration (pass) is in flight per pipe, each on its own HIP stream.
struct Pipe
{
    hipStream_t stream = nullptr;
    // path state by queue slot, two generations (FrameView::gen); shadow rays by shadow-queue slot
    float4 *   rayO[2] = {nullptr, nullptr}, *rayD[2] = {nullptr, nullptr}, *hit[2] = {nullptr, nullptr},
           *thr[2] = {nullptr, nullptr};
    float4 *   shO = nullptr, *shD = nullptr, *eiVis = nullptr, *contrib = nullptr;
    float2*    aaOff  = nullptr;
    uint32_t*  hitIds = nullptr;
    uint8_t *  dead = nullptr, *nvert = nullptr;
    uint32_t * q[2] = {nullptr, nullptr}, *qShadow = nullptr;
    uint32_t*  counters   = nullptr;  // kCounterWords per pass
    uint32_t   contribCap = 0;        // path vertices the contrib buffer holds
    size_t     cap        = 0;        // pixels (image pixels x batched iterations) the per-pixel buffers hold
    uint32_t   lastSub    = 1;        // iterations in the pipe's last pass
    hipEvent_t evDone = nullptr, evFree = nullptr;
    bool       allocated = false, busy = false;  // busy: a resolve that frees the pipe has been enqueued
};

// A binary build tree over N primitives (the triangles of a range of the scene, or the instance boxes of the top
// level) and what its traversal layout needs. The build nodes and the order stay for refits and inspection.
struct Tree
{
    BuildNode* dBuildNodes = nullptr;
    uint32_t*  dOrder      = nullptr;
    uint32_t   first = 0, N = 0;  // triangle range (N instances for the top level)
    uint32_t   nNodes = 0, nInner = 0, depth = 0;
    LevelTable lt{};
    QuantGrid  grid{};
    float      sceneAbs = 0.0f;
    float      bmin[3] = {0.0f, 0.0f, 0.0f}, bmax[3] = {0.0f, 0.0f, 0.0f};
    NodePlace  place{0, 0, 0, 0};  // position in the node / leaf arrays shared by all structures (prt_bvh.hpp)
    int32_t    root = 0;
    uint32_t*  dWideIdx = nullptr;  // between tree_layout and tree_emit: BVH4 index of every build node
};

struct prt_context
{
    int         device = 0;
    hipStream_t stream = nullptr;  // accumulation order: k_resolve of every pass, scene upload, BVH build, read-backs
    uint32_t    w = 0, h = 0, npix = 0;
    uint32_t    flags = 0, seed = 1;
    int         numCU = 256;
    uint32_t    gridExtend = 0, gridShadow = 0, gridBatch = 0, gridAo = 0, gridFused = 0, gridShade = 0;
    bool        fuse = true;  // shadow(b) + extend(b+1) in one launch (PRT_FUSE=0 turns it off)
    uint32_t    batchFit = 0;      // iterations per pass that the last attempt could allocate (0 = not yet limited)
    uint32_t    batchUsed = 0;     // iterations per pass of the last batched pass (prt_get_pass_iterations)
    uint64_t    passBytesMax = 0;  // memory policy: upper bound of the per-pass buffers (0 = none), PRT_MAX_PASS_BYTES
    bool        oom = false;       // the last pipe allocation failed for lack of memory
    uint32_t    batch = 1;    // iterations per pass in prt_render (see there)  // resident blocks of the persistent kernels
    std::string err;

    // scene
    prt_vertex*   dVerts = nullptr;
    uint32_t*     dIdx   = nullptr;
    prt_material* dMats  = nullptr;
    TexDesc*      dTex   = nullptr;
    float4*       dShadeTris = nullptr;  // 7 x float4 per triangle (k_gather_shading)
    std::vector<void*> texMem;
    uint32_t      nverts = 0, ntri = 0, nmats = 0, ntex = 0;
    bool          haveScene = false, haveAccel = false;
    bool          accelStale = false;  // vertices replaced since the last build / refit

    // acceleration structure. Single-level: `top` is the tree over all triangles. Two-level (prt_set_instances):
    // `top` is the tree over the instances' world boxes, `blas` the trees of the bottom-level structures; all of them
    // are laid out in the one node array / leaf array the kernels see.
    Tree              top;
    std::vector<Tree> blas;
    uint4*     dNodes      = nullptr;  // BVH4, 4 x uint4 per node
    float4*    dTris       = nullptr;
    size_t     nodesCap = 0, trisCap = 0;  // nodes / 64-byte leaf slots allocated: the arrays stay when a refit fits
    size_t     instCap = 0, instBaseCap = 0;  // float4 of dInst (dShadeInst has as many) / words of dInstBase allocated
    uint32_t   emittedTopHot = 0xffffffffu;  // two-level: hot nodes the top level had when the bottom level was emitted
    uint32_t   topColdBase   = 0;            // two-level: where the top level's other nodes start (end of the node array)
    float      buildMs = 0.0f, refitMs = 0.0f;
    // instances (two-level scenes)
    std::vector<prt_blas_range> blasRanges;
    std::vector<prt_instance>   instances;
    float4*   dInst      = nullptr;  // traversal records, top-level leaf order (prt_trace.hpp InstanceRec)
    float4*   dShadeInst = nullptr;  // shading records, caller's order (prt_kernels.hpp ShadeInst)
    uint32_t* dInstBase  = nullptr;  // flattened primitive id of every instance's first triangle, + total; then the bucket table
    uint32_t  instShift  = 0;        // bucket of a flattened primitive id = id >> instShift (instance_tables)
    float     padScale   = 1.0f;
    uint32_t  nHot       = 0;  // nodes 0 .. nHot-1 of dNodes are the tops of the trees (BvhView::nTop)
    bool      twoLevel() const { return !instances.empty(); }
    uint32_t  gridTwo[6] = {0, 0, 0, 0, 0, 0};  // resident grids of the two-level kernel variants

    // sampling
    uint32_t* dSobol    = nullptr;
    uint32_t* dSobolBytes = nullptr;  // [64][4][256] byte tables of the same matrices
    uint32_t* dScramble = nullptr;

    // frame
    prt_uniforms        ubo{};
    float4 *            accumOwn = nullptr, *accum = nullptr;
    Pipe                pipes[kMaxPipes];
    int                 nPipes   = 2;   // passes in flight (prt_set_pipeline_depth)
    int                 nextPipe = 0, lastPipe = 0;
    int                 lastMode = -1;
    bool                lastPipelined = false;
    uint32_t*           batchCounter = nullptr;
    unsigned long long* stats        = nullptr;
    uint64_t            iterations   = 0;

    // profiling
    int profLevel = 0;
    uint32_t    profMask = 0xffffffffu;  // kernel kinds that get events (prt_set_profiling_mask)
    struct Pending
    {
        int        kind;
        hipEvent_t e0, e1;
    };
    std::vector<Pending>    pending;
    std::vector<hipEvent_t> eventPool;
    double                  profMs[PRT_K_COUNT]       = {};
    uint64_t                profLaunches[PRT_K_COUNT] = {};
};

static int fail(prt_context* c, int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if(c)
        c->err = buf;
    else
        g_createError = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                            \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (expr);                                                                    \
        if(e_ != hipSuccess)                                                                       \
            return fail(c, PRT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while(0)

template <class T>
static hipError_t dalloc(T*& p, size_t n)
{
    return hipMalloc((void**)&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
}
template <class T>
static void dfree(T*& p)
{
    if(p) (void)hipFree(p);
    p = nullptr;
}
static inline uint32_t cdiv(size_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

static void pipe_free_buffers(Pipe& p);
static void pipe_free(Pipe& p)
{
    pipe_free_buffers(p);
    dfree(p.counters);
    if(p.evDone) (void)hipEventDestroy(p.evDone);
    if(p.evFree) (void)hipEventDestroy(p.evFree);
    if(p.stream) (void)hipStreamDestroy(p.stream);
    p = Pipe();
}
static void pipe_free_buffers(Pipe& p)
{
    for(int g = 0; g < 2; ++g)
    {
        dfree(p.rayO[g]); dfree(p.rayD[g]); dfree(p.hit[g]); dfree(p.thr[g]);
    }
    dfree(p.shO); dfree(p.shD); dfree(p.eiVis);
    dfree(p.contrib); dfree(p.aaOff); dfree(p.hitIds); dfree(p.dead); dfree(p.nvert); dfree(p.q[0]); dfree(p.q[1]);
    dfree(p.qShadow);
    p.contribCap = 0;
    p.cap        = 0;
}
// buffers for n pixels (grows: the caller has made the pipe idle)
static hipError_t pipe_alloc(Pipe& p, size_t n)
{
    hipError_t e = hipSuccess;
    if(!p.allocated)
    {
        e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        if(e == hipSuccess) e = hipEventCreateWithFlags(&p.evDone, hipEventDisableTiming);
        if(e == hipSuccess) e = hipEventCreateWithFlags(&p.evFree, hipEventDisableTiming);
        if(e == hipSuccess) e = dalloc(p.counters, (size_t)kCounterWords);
        if(e == hipSuccess) p.allocated = true;
    }
    if(e != hipSuccess || n <= p.cap) return e;
    pipe_free_buffers(p);
    for(int g = 0; g < 2; ++g)
    {
        if(e == hipSuccess) e = dalloc(p.rayO[g], n);
        if(e == hipSuccess) e = dalloc(p.rayD[g], n);
        if(e == hipSuccess) e = dalloc(p.hit[g], n);
        if(e == hipSuccess) e = dalloc(p.thr[g], n);
    }
    if(e == hipSuccess) e = dalloc(p.shO, n);
    if(e == hipSuccess) e = dalloc(p.shD, n);
    if(e == hipSuccess) e = dalloc(p.eiVis, n);
    if(e == hipSuccess) e = dalloc(p.aaOff, n);
    if(e == hipSuccess) e = dalloc(p.hitIds, n);
    if(e == hipSuccess) e = dalloc(p.dead, n);
    if(e == hipSuccess) e = dalloc(p.nvert, n);
    if(e == hipSuccess) e = dalloc(p.q[0], n);
    if(e == hipSuccess) e = dalloc(p.q[1], n);
    if(e == hipSuccess) e = dalloc(p.qShadow, n);
    // in stream order with the pipe's kernels: hipMemset on the null stream may return before the fill has run, and
    // a non-blocking stream does not wait for it
    if(e == hipSuccess) e = hipMemsetAsync(p.dead, 0, n, p.stream);
    if(e == hipSuccess) e = hipMemsetAsync(p.nvert, 0, n, p.stream);
    if(e == hipSuccess) e = hipMemsetAsync(p.hitIds, 0xff, n * 4, p.stream);
    if(e == hipSuccess) e = hipStreamSynchronize(p.stream);
    if(e == hipSuccess) p.cap = n;
    return e;
}
// every stream of the context idle
static hipError_t sync_all(prt_context* c)
{
    hipError_t e = hipSuccess;
    for(Pipe& p : c->pipes)
        if(p.stream)
        {
            const hipError_t e2 = hipStreamSynchronize(p.stream);
            if(e == hipSuccess) e = e2;
            p.busy = false;
        }
    if(c->stream)
    {
        const hipError_t e2 = hipStreamSynchronize(c->stream);
        if(e == hipSuccess) e = e2;
    }
    return e;
}

static void prof_collect(prt_context* c);

// traversal knobs; PRT_TUNE="refillMin,stallMax,leafMin" overrides (measurement only)
static TraceTune trace_tune()
{
    static TraceTune t = [] {
        TraceTune v{16u, 28u, 16u, 0u};  // tools/tune_sweep.sh on the bench workload
        if(const char* e = getenv("PRT_TUNE"))
        {
            unsigned a = v.refillMin, b = v.stallMax, d = v.leafMin;
            if(sscanf(e, "%u,%u,%u", &a, &b, &d) >= 1)
            {
                v.refillMin = a < 1 ? 1 : (a > 64 ? 64 : a);
                v.stallMax  = b > 64 ? 64 : b;
                v.leafMin   = d < 1 ? 1 : (d > 64 ? 64 : d);
            }
        }
        return v;
    }();
    return t;
}

static void tree_free(Tree& t)
{
    dfree(t.dBuildNodes);
    dfree(t.dOrder);
    dfree(t.dWideIdx);
    t = Tree();
}
static void free_accel(prt_context* c)
{
    tree_free(c->top);
    for(Tree& t : c->blas) tree_free(t);
    c->blas.clear();
    dfree(c->dNodes);
    dfree(c->dTris);
    c->nodesCap = c->trisCap = 0;
    c->emittedTopHot = 0xffffffffu;
    dfree(c->dInst);
    dfree(c->dShadeInst);
    dfree(c->dInstBase);
    c->instCap = c->instBaseCap = 0;
    c->haveAccel = false;
}
static void free_scene(prt_context* c)
{
    dfree(c->dVerts);
    dfree(c->dIdx);
    dfree(c->dMats);
    dfree(c->dTex);
    dfree(c->dShadeTris);
    for(void* p : c->texMem) (void)hipFree(p);
    c->texMem.clear();
    free_accel(c);
    c->blasRanges.clear();
    c->instances.clear();
    c->haveScene = false;
}

// The traversal kernels exist in four instantiations each, <INSTR, TWO>: with the counters of profiling level 2 or
// without, for a two-level scene or a single-level one. One table per kernel, indexed [instrumented][two-level]; the
// resident grid of a two-level variant is gridTwo[...], of a single-level one the kernel's own.
template <class K>
struct KernelTable
{
    K fn[2][2];
    K pick(const prt_context* c) const { return fn[c->profLevel >= 2 ? 1 : 0][c->twoLevel() ? 1 : 0]; }
};
#define PRT_KERNEL_TABLE(k) {{{k<false, false>, k<false, true>}, {k<true, false>, k<true, true>}}}
static const KernelTable<decltype(&k_extend<false, false>)>        kExtendTable        = PRT_KERNEL_TABLE(k_extend);
static const KernelTable<decltype(&k_shadow<false, false>)>        kShadowTable        = PRT_KERNEL_TABLE(k_shadow);
static const KernelTable<decltype(&k_shadow_extend<false, false>)> kShadowExtendTable  = PRT_KERNEL_TABLE(k_shadow_extend);
#undef PRT_KERNEL_TABLE

extern "C" {

const char* prt_last_error(const prt_context* c) { return c ? c->err.c_str() : g_createError.c_str(); }

int prt_create(int device, uint32_t width, uint32_t height, prt_context** out)
{
    if(!out || width == 0 || height == 0) return fail(nullptr, PRT_E_INVALID, "prt_create: bad arguments");
    if((uint64_t)width * height >= (1ull << 27))
        return fail(nullptr, PRT_E_LIMIT, "prt_create: more than 2^27 pixels");
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, PRT_E_HIP, "prt_create: no HIP device available (the MI355X path has no CPU fallback)");
    if(device < 0 || device >= ndev) return fail(nullptr, PRT_E_INVALID, "prt_create: device %d of %d", device, ndev);
    prt_context* c = new prt_context();
    c->device      = device;
    c->w           = width;
    c->h           = height;
    c->npix        = width * height;
    hipError_t e   = hipSetDevice(device);
    if(e == hipSuccess) e = scratch_for_device(device);  // set-up of prt_trace_device_noalloc, once per GPU
    if(e == hipSuccess) e = hipStreamCreate(&c->stream);
    hipDeviceProp_t prop;
    if(e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if(e == hipSuccess) c->numCU = prop.multiProcessorCount;
    if(e == hipSuccess)
    {
        // persistent kernels: exactly as many blocks as stay resident
        auto resident = [&](const void* fn) -> uint32_t {
            int nb = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, PRT_TRACE_BLOCK, 0) != hipSuccess || nb < 1) nb = 4;
            return (uint32_t)nb * (uint32_t)c->numCU;
        };
        c->gridExtend = resident((const void*)k_extend<false>);
        c->gridShadow = resident((const void*)k_shadow<false>);
        c->gridBatch  = resident((const void*)k_trace_batch<false>);
        c->gridAo     = resident((const void*)k_ao_shade<false>);
        c->gridFused  = resident((const void*)k_shadow_extend<false>);
        // k_shade strides over its queue with a fixed stride: every block must be resident from the start, or the
        // late blocks do their whole share on a nearly empty GPU
        {
            int nb = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_shade<false>, PRT_SHADE_BLOCK, 0) != hipSuccess || nb < 1) nb = 4;
            c->gridShade = (uint32_t)nb * (uint32_t)c->numCU;
        }
        if(const char* g = getenv("PRT_SHADE_BLOCKS_PER_CU"))
        {
            const int nb = atoi(g);
            if(nb >= 1 && nb <= 16) c->gridShade = (uint32_t)nb * (uint32_t)c->numCU;
        }
        if(const char* fe = getenv("PRT_FUSE")) c->fuse = atoi(fe) != 0;
        // default batch: enough iterations to put ~128 M pixels into a pass (64 at 1080p, 16 at 4K), at most 64: a launch's
        // tail (its longest rays on an emptying GPU) is paid once per pass. Measured at 1080p, Mrays/s with one pass in
        // flight: 2: 5790, 4: 6840, 8: 7530, 16: 7920, 32: 8170, 64: 8280 (38 GB of path state at 64, on one pipe: see
        // prt_render).
        {
            const uint64_t target = 128ull << 20;
            uint64_t       b      = (target + c->npix / 2) / c->npix;
            c->batch              = (uint32_t)(b < 1 ? 1 : (b > 64 ? 64 : b));
            if(const char* be = getenv("PRT_BATCH"))
            {
                const int v = atoi(be);
                if(v >= 1 && v <= 64) c->batch = (uint32_t)v;
            }
            if(const char* me = getenv("PRT_MAX_PASS_BYTES")) c->passBytesMax = strtoull(me, nullptr, 10);
        }
        if(const char* g = getenv("PRT_GRID_BLOCKS_PER_CU"))
        {
            const int nb = atoi(g);
            if(nb >= 1 && nb <= 16) c->gridExtend = c->gridShadow = c->gridBatch = c->gridAo = c->gridFused = (uint32_t)nb * (uint32_t)c->numCU;
        }
    }
    const size_t n = c->npix;
    if(e == hipSuccess) e = dalloc(c->accumOwn, n);
    if(e == hipSuccess) e = pipe_alloc(c->pipes[0], n);
    if(e == hipSuccess) e = dalloc(c->batchCounter, (size_t)4);
    if(e == hipSuccess) e = dalloc(c->stats, (size_t)kStatWords);
    if(e == hipSuccess) e = dalloc(c->dSobol, (size_t)64 * 32);
    if(e == hipSuccess)
    {
        uint32_t tbl[64 * 32];
        build_sobol_matrices(tbl);
        e = hipMemcpy(c->dSobol, tbl, sizeof tbl, hipMemcpyHostToDevice);
        std::vector<uint32_t> bytes((size_t)64 * 4 * 256);
        for(int d = 0; d < 64; ++d)
            for(int b = 0; b < 4; ++b)
                for(int v = 0; v < 256; ++v)
                {
                    uint32_t x = 0;
                    for(int j = 0; j < 8; ++j)
                        if(v & (1 << j)) x ^= tbl[d * 32 + 8 * b + j];
                    bytes[((size_t)d * 4 + b) * 256 + v] = x;
                }
        if(e == hipSuccess) e = dalloc(c->dSobolBytes, bytes.size());
        if(e == hipSuccess) e = hipMemcpy(c->dSobolBytes, bytes.data(), bytes.size() * 4, hipMemcpyHostToDevice);
    }
    // stream-ordered fills (see pipe_alloc), finished before the context is handed out
    if(e == hipSuccess) e = hipMemsetAsync(c->accumOwn, 0, n * sizeof(float4), c->stream);
    if(e == hipSuccess) e = hipMemsetAsync(c->stats, 0, kStatWords * sizeof(unsigned long long), c->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
    // each pipe wants a hardware queue of its own next to the main stream (and the host application's streams)
    if(const char* qe = getenv("GPU_MAX_HW_QUEUES"))
        if(atoi(qe) >= 8) c->nPipes = 3;
    if(const char* pe = getenv("PRT_PIPES"))
    {
        const int np = atoi(pe);
        if(np >= 1 && np <= kMaxPipes) c->nPipes = np;
    }
    if(e != hipSuccess)
    {
        fail(nullptr, PRT_E_HIP, "prt_create: %s", hipGetErrorString(e));
        prt_destroy(c);
        return PRT_E_HIP;
    }
    c->accum = c->accumOwn;
    // default uniforms = vkContext.h:54-82 defaults
    memset(&c->ubo, 0, sizeof c->ubo);
    c->ubo.numIndirectBounces = 4;
    c->ubo.samplesPerPixel    = 1;
    c->ubo.numAArays          = 1;
    c->ubo.filterRadius       = 1.0f;
    c->ubo.numAOrays          = 16;
    c->ubo.aoRayLength        = 1.0f;
    *out = c;
    return PRT_OK;
}

int prt_destroy(prt_context* c)
{
    if(!c) return PRT_OK;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    free_scene(c);
    dfree(c->accumOwn);
    dfree(c->batchCounter);
    dfree(c->stats);
    dfree(c->dSobol);
    dfree(c->dSobolBytes);
    dfree(c->dScramble);
    prof_collect(c);
    for(hipEvent_t e : c->eventPool) (void)hipEventDestroy(e);
    for(Pipe& p : c->pipes) pipe_free(p);
    if(c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PRT_OK;
}

int prt_set_scene(prt_context* c, const prt_vertex* v, size_t nv, const uint32_t* idx, size_t nidx,
                  const prt_material* m, size_t nm, const prt_texture* tex, size_t ntex)
{
    if(!c) return PRT_E_INVALID;
    if(!v || !idx || !m || nv == 0 || nidx < 3 || nm == 0 || (ntex && !tex))
        return fail(c, PRT_E_INVALID, "prt_set_scene: empty vertex/index/material input");
    if(nidx / 3 >= (1u << 28)) return fail(c, PRT_E_LIMIT, "prt_set_scene: more than 2^28 triangles");
    for(size_t i = 0; i < (nidx / 3) * 3; ++i)
        if(idx[i] >= nv) return fail(c, PRT_E_INVALID, "prt_set_scene: index %zu (= %u) out of range", i, idx[i]);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    free_scene(c);
    c->nverts = (uint32_t)nv;
    c->ntri   = (uint32_t)(nidx / 3);
    c->nmats  = (uint32_t)nm;
    c->ntex   = (uint32_t)ntex;
    HIPCHK(c, dalloc(c->dVerts, nv));
    HIPCHK(c, dalloc(c->dIdx, (size_t)c->ntri * 3));
    HIPCHK(c, dalloc(c->dMats, nm));
    HIPCHK(c, hipMemcpy(c->dVerts, v, nv * sizeof(prt_vertex), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dIdx, idx, (size_t)c->ntri * 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dMats, m, nm * sizeof(prt_material), hipMemcpyHostToDevice));
    std::vector<TexDesc> td(ntex);
    for(size_t i = 0; i < ntex; ++i)
    {
        if(!tex[i].rgba || tex[i].width == 0 || tex[i].height == 0)
            return fail(c, PRT_E_INVALID, "prt_set_scene: texture %zu is empty", i);
        const size_t bytes = (size_t)tex[i].width * tex[i].height * 4;
        void*        p     = nullptr;
        HIPCHK(c, hipMalloc(&p, bytes));
        c->texMem.push_back(p);
        HIPCHK(c, hipMemcpy(p, tex[i].rgba, bytes, hipMemcpyHostToDevice));
        td[i].rgba = (const uint8_t*)p;
        td[i].w    = tex[i].width;
        td[i].h    = tex[i].height;
    }
    HIPCHK(c, dalloc(c->dTex, ntex));
    if(ntex) HIPCHK(c, hipMemcpy(c->dTex, td.data(), ntex * sizeof(TexDesc), hipMemcpyHostToDevice));
    HIPCHK(c, dalloc(c->dShadeTris, (size_t)c->ntri * PRT_SHADE_TRI_F4));
    hipLaunchKernelGGL(k_gather_shading, dim3(cdiv(c->ntri, 256)), dim3(256), 0, c->stream, c->dVerts, c->dIdx, c->ntri,
                       c->dShadeTris);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->haveScene = true;
    return PRT_OK;
}

static int device_scan(prt_context* c, uint32_t* data, uint32_t n, uint32_t* tileSums, uint32_t* totalDev)
{
    const uint32_t tiles = cdiv(n, PRT_SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(PRT_SCAN_BLOCK), 0, c->stream, data, n, tileSums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(PRT_SCAN_BLOCK), 0, c->stream, tileSums, tiles, totalDev);
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(PRT_SCAN_BLOCK), 0, c->stream, data, n, tileSums);
    HIPCHK(c, hipGetLastError());
    return PRT_OK;
}

#define TCHK(expr)                                                                                          \
    do                                                                                                      \
    {                                                                                                       \
        hipError_t e_ = (expr);                                                                             \
        if(e_ != hipSuccess)                                                                                \
        {                                                                                                   \
            cleanup();                                                                                      \
            return fail(c, PRT_E_HIP, "acceleration structure: %s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                   \
    } while(0)

// Breadth-first binned-SAH build (prt_bvh.hpp) of one tree over t.N primitives: the triangles t.first .. t.first+N-1
// of the scene, or, with `given`, N boxes (2 x float4 each: the instances of a two-level scene). Leaves hold at most
// leafMax primitives. On success t holds build nodes, order, level table and depth.
static int tree_build(prt_context* c, Tree& t, const float4* given, uint32_t leafMax, uint32_t maxDepth)
{
    const uint32_t  N   = t.N;
    hipStream_t     st  = c->stream;
    const uint32_t* idx = c->dIdx + (size_t)3 * t.first;
    dfree(t.dBuildNodes);
    dfree(t.dOrder);
    dfree(t.dWideIdx);
    const uint32_t maxSplit = N / (leafMax + 1) + 1;
    uint32_t *     order2 = nullptr, *nodeOf = nullptr, *nodeOf2 = nullptr, *cb = nullptr, *bins = nullptr;
    uint32_t *     flags = nullptr, *scanBuf = nullptr, *tileSums = nullptr, *rank = nullptr, *splitOf = nullptr;
    uint32_t*      total = nullptr;
    float4*        pbox  = nullptr;
    SplitRec*      splits = nullptr;
    int            rc     = PRT_OK;
    auto           cleanup = [&]() {
        dfree(order2); dfree(nodeOf); dfree(nodeOf2); dfree(cb); dfree(bins); dfree(flags); dfree(scanBuf);
        dfree(tileSums); dfree(rank); dfree(splitOf); dfree(total); dfree(pbox); dfree(splits);
    };
    TCHK(dalloc(t.dBuildNodes, (size_t)2 * N));
    TCHK(dalloc(t.dOrder, (size_t)N));
    TCHK(dalloc(order2, (size_t)N));
    TCHK(dalloc(nodeOf, (size_t)N));
    TCHK(dalloc(nodeOf2, (size_t)N));
    TCHK(dalloc(cb, (size_t)12 * N));
    TCHK(dalloc(bins, (size_t)maxSplit * kBinWords));
    TCHK(dalloc(flags, (size_t)2 * N));
    TCHK(dalloc(scanBuf, (size_t)2 * N));
    TCHK(dalloc(tileSums, (size_t)cdiv((size_t)2 * N, PRT_SCAN_TILE) + 1));
    TCHK(dalloc(rank, (size_t)N));
    TCHK(dalloc(splitOf, (size_t)N));
    TCHK(dalloc(total, (size_t)1));
    TCHK(dalloc(pbox, (size_t)2 * N));
    TCHK(dalloc(splits, (size_t)maxSplit));

    uint32_t* orderCur = t.dOrder;
    uint32_t* orderAlt = order2;
    hipLaunchKernelGGL(k_build_reset, dim3(1), dim3(64), 0, st, t.dBuildNodes, cb, N);
    hipLaunchKernelGGL(k_prim_boxes, dim3(cdiv(N, 256)), dim3(256), 0, st, c->dVerts, idx, N, pbox, orderCur, nodeOf,
                       t.dBuildNodes, cb, given);
    uint32_t   levelStart = 0, levelCount = 1, nNodes = 1, depth = 1;
    LevelTable lt;
    memset(&lt, 0, sizeof lt);
    for(;;)
    {
        lt.start[lt.n++] = levelStart;
        hipLaunchKernelGGL(k_level_flags, dim3(cdiv(levelCount, 256)), dim3(256), 0, st, t.dBuildNodes, levelStart,
                           levelCount, rank, leafMax);
        rc = device_scan(c, rank, levelCount, tileSums, total);
        if(rc) { cleanup(); return rc; }
        uint32_t nsplit = 0;
        TCHK(hipMemcpyAsync(&nsplit, total, 4, hipMemcpyDeviceToHost, st));
        TCHK(hipStreamSynchronize(st));
        if(nsplit == 0) break;
        if(nsplit > maxSplit) { cleanup(); return fail(c, PRT_E_LIMIT, "prt_build_accel: split overflow"); }
        hipLaunchKernelGGL(k_level_prepare, dim3(cdiv(levelCount, 256)), dim3(256), 0, st, t.dBuildNodes, levelStart,
                           levelCount, rank, splits, leafMax);
        hipLaunchKernelGGL(k_split_of, dim3(cdiv(levelCount, 256)), dim3(256), 0, st, t.dBuildNodes, levelStart,
                           levelCount, rank, splitOf, leafMax);
        hipLaunchKernelGGL(k_clear_bins, dim3(cdiv((size_t)nsplit * kBinWords, 256)), dim3(256), 0, st, bins, nsplit);
        hipLaunchKernelGGL(k_bin, dim3(cdiv(N, 256)), dim3(256), 0, st, orderCur, nodeOf, t.dBuildNodes, cb, pbox, rank,
                           levelStart, N, bins, leafMax);
        hipLaunchKernelGGL(k_split, dim3(cdiv(nsplit, 64)), dim3(64), 0, st, t.dBuildNodes, cb, bins, splits, nsplit,
                           nNodes);
        hipLaunchKernelGGL(k_flags, dim3(cdiv(N, 256)), dim3(256), 0, st, orderCur, nodeOf, splitOf, splits,
                           t.dBuildNodes, pbox, levelStart, N, flags, scanBuf);
        rc = device_scan(c, scanBuf, N, tileSums, total);
        if(rc) { cleanup(); return rc; }
        hipLaunchKernelGGL(k_scatter, dim3(cdiv(N, 256)), dim3(256), 0, st, orderCur, nodeOf, splitOf, splits,
                           t.dBuildNodes, cb, pbox, flags, scanBuf, levelStart, N, orderAlt, nodeOf2);
        TCHK(hipGetLastError());
        std::swap(orderCur, orderAlt);
        std::swap(nodeOf, nodeOf2);
        levelStart = nNodes;
        levelCount = 2 * nsplit;
        nNodes += 2 * nsplit;
        depth++;
        if(depth > maxDepth)
        {
            cleanup();
            return fail(c, PRT_E_LIMIT, "prt_build_accel: tree deeper than %u levels", maxDepth);
        }
    }
    if(orderCur != t.dOrder)
        TCHK(hipMemcpyAsync(t.dOrder, orderCur, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    TCHK(hipGetLastError());
    TCHK(hipStreamSynchronize(st));
    cleanup();
    t.lt     = lt;
    t.nNodes = nNodes;
    t.depth  = depth;
    return PRT_OK;
}

// What the traversal layout of a tree needs before it can be emitted: the root box (-> quantisation grid: BVH4 with
// 16-bit boxes on the root box, see prt_bvh.hpp), and the BVH4 index of every build node (-> number of wide nodes).
static int tree_layout(prt_context* c, Tree& t)
{
    hipStream_t st = c->stream;
    uint32_t *  tileSums = nullptr, *total = nullptr;
    dfree(t.dWideIdx);
    auto cleanup = [&]() { dfree(tileSums); dfree(total); };
    TCHK(dalloc(t.dWideIdx, (size_t)t.nNodes));
    TCHK(dalloc(tileSums, (size_t)cdiv((size_t)t.nNodes, PRT_SCAN_TILE) + 1));
    TCHK(dalloc(total, (size_t)1));
    BuildNode rootNode;
    TCHK(hipMemcpyAsync(&rootNode, t.dBuildNodes, sizeof rootNode, hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    auto dec = [](uint32_t k) {
        const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
        float          f;
        memcpy(&f, &u, 4);
        return f;
    };
    float sa = 0.0f;
    for(int k = 0; k < 3; ++k)
    {
        const float lo = dec(rootNode.bmin[k]), hi = dec(rootNode.bmax[k]);
        sa             = sa < fabsf(lo) ? fabsf(lo) : sa;
        sa             = sa < fabsf(hi) ? fabsf(hi) : sa;
        const float ext = hi - lo;
        t.bmin[k]        = lo;
        t.bmax[k]        = hi;
        t.grid.qmin[k]   = lo;
        t.grid.qscale[k] = ext > 0.0f ? (ext / 65535.0f) * 1.0001f : 1.0f;  // qmin + 65535*scale >= hi
    }
    t.sceneAbs = sa;
    hipLaunchKernelGGL(k_wide_flags, dim3(cdiv(t.nNodes, 256)), dim3(256), 0, st, t.dBuildNodes, t.nNodes, t.lt,
                       t.dWideIdx);
    const int rc = device_scan(c, t.dWideIdx, t.nNodes, tileSums, total);
    if(rc) { cleanup(); return rc; }
    uint32_t nInner = 0;
    TCHK(hipMemcpyAsync(&nInner, total, 4, hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    cleanup();
    t.nInner = nInner;
    return PRT_OK;
}

// BVH4 nodes (and, for trees over triangles, the leaf records) of one tree at its place in the shared arrays
static void tree_emit(prt_context* c, Tree& t, bool triangles)
{
    hipStream_t st = c->stream;
    const uint32_t* idx = c->dIdx + (size_t)3 * t.first;
    hipLaunchKernelGGL(k_emit_wide, dim3(cdiv(t.nNodes, 256)), dim3(256), 0, st, t.dBuildNodes, t.nNodes, t.lt,
                       t.dWideIdx, t.grid, c->dNodes, t.place, triangles ? c->dVerts : (const prt_vertex*)nullptr, idx,
                       t.dOrder);
    if(triangles)
        hipLaunchKernelGGL(k_emit_leaves, dim3(cdiv(t.nNodes, 256)), dim3(256), 0, st, t.dBuildNodes, t.nNodes,
                           c->dVerts, idx, t.dOrder, c->dTris, t.place.slotBase, t.nInner ? 1u : 0u);
    // wide node 0 is the root: first of the hot part when the tree has one, else first of the cold part
    t.root = t.nInner ? (int32_t)(t.place.hotCount ? t.place.hotBase : t.place.coldBase) : (int32_t) ~((t.place.slotBase << 3) | t.N);
}

static bool is_identity(const float* m)
{
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return memcmp(m, id, sizeof id) == 0;
}
static prt_host::mat4 to_mat4(const float* m)
{
    prt_host::mat4 r;
    for(int cc = 0; cc < 4; ++cc)
        for(int rr = 0; rr < 4; ++rr) r[cc][rr] = m[cc * 4 + rr];
    return r;
}

// kappa = |M|inf * |inverse(M)|inf of the 3 x 3 part (>= 1); false when the transform is singular, not finite or
// conditioned worse than 1024 (the widening of the top-level box test is bounded with it)
static bool instance_condition(const float* M, double& kappa)
{
    kappa = 1.0;
    for(int k = 0; k < 16; ++k)
        if(!std::isfinite(M[k])) { kappa = NAN; return false; }
    if(is_identity(M)) return true;
    const prt_host::mat4 W = prt_host::inverse(to_mat4(M));
    double nM = 0.0, nW = 0.0;
    bool   finite = true;
    for(int r = 0; r < 3; ++r)
    {
        const double rm = fabs((double)M[0 + r]) + fabs((double)M[4 + r]) + fabs((double)M[8 + r]);
        const double rw = fabs((double)W[0][r]) + fabs((double)W[1][r]) + fabs((double)W[2][r]);
        nM = rm > nM ? rm : nM;
        nW = rw > nW ? rw : nW;
    }
    for(int cc = 0; cc < 4; ++cc)
        for(int rr = 0; rr < 4; ++rr) finite = finite && std::isfinite(W[cc][rr]);
    kappa = nM * nW;
    if(!finite || !(kappa == kappa) || kappa > 1024.0) return false;
    if(kappa < 1.0) kappa = 1.0;
    return true;
}

// Top level of a two-level scene: conservative world boxes of the instances (bottom-level root box through the
// transform, in double, rounded outwards and padded), a tree over them with one instance per leaf.
static int tlas_build(prt_context* c)
{
    const size_t        n = c->instances.size();
    std::vector<float4> boxes(2 * n);
    double              padScale = 1.0;
    for(size_t i = 0; i < n; ++i)
    {
        const prt_instance& in = c->instances[i];
        const Tree&         b  = c->blas[in.blas];
        const float*        M  = in.transform;
        double              lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, amax = 0.0;
        for(int cr = 0; cr < 8; ++cr)
        {
            const double p[3] = {(cr & 1) ? b.bmax[0] : b.bmin[0], (cr & 2) ? b.bmax[1] : b.bmin[1],
                                 (cr & 4) ? b.bmax[2] : b.bmin[2]};
            for(int k = 0; k < 3; ++k)
            {
                const double w = (double)M[0 + k] * p[0] + (double)M[4 + k] * p[1] + (double)M[8 + k] * p[2] + (double)M[12 + k];
                lo[k] = w < lo[k] ? w : lo[k];
                hi[k] = w > hi[k] ? w : hi[k];
                amax  = fabs(w) > amax ? fabs(w) : amax;
            }
        }
        // condition of the linear part (row-sum norms): scales the rounding error of o' = W*o, M*o' - o
        double kappa = 1.0;
        if(!instance_condition(M, kappa) || !(amax < 1e30))
            return fail(c, PRT_E_INVALID, "prt_build_accel: the transform of instance %zu is singular or too ill-conditioned (condition %g)", i, kappa);
        padScale = kappa * kappa > padScale ? kappa * kappa : padScale;
        const double pad = 16.0 * 5.9604644775390625e-08 * kappa * amax + 1e-30;
        for(int k = 0; k < 3; ++k)
        {
            ((float*)&boxes[2 * i])[k]     = nextafterf((float)(lo[k] - pad), -INFINITY);
            ((float*)&boxes[2 * i + 1])[k] = nextafterf((float)(hi[k] + pad), INFINITY);
        }
        boxes[2 * i].w = boxes[2 * i + 1].w = 0.0f;
    }
    c->padScale = (float)padScale;
    float4* dBoxes = nullptr;
    HIPCHK(c, dalloc(dBoxes, 2 * n));
    hipError_t e = hipMemcpy(dBoxes, boxes.data(), 2 * n * sizeof(float4), hipMemcpyHostToDevice);
    int        rc = PRT_OK;
    if(e == hipSuccess)
    {
        c->top.first = 0;
        c->top.N     = (uint32_t)n;
        rc           = tree_build(c, c->top, dBoxes, 1u, PRT_TLAS_MAX_DEPTH);
    }
    dfree(dBoxes);
    HIPCHK(c, e);
    return rc;
}

// Instance tables of a two-level scene: traversal records in top-level leaf order, shading records + primitive bases in
// the caller's order. World -> object = glm's inverse(mat4) (host/prt_math.h), as the oracle computes it.
static int instance_tables(prt_context* c)
{
    const size_t          n = c->instances.size();
    std::vector<uint32_t> order(n), base(n + 1);
    HIPCHK(c, hipMemcpy(order.data(), c->top.dOrder, n * 4, hipMemcpyDeviceToHost));
    uint64_t b = 0;
    for(size_t i = 0; i < n; ++i)
    {
        base[i] = (uint32_t)b;
        b += c->blas[c->instances[i].blas].N;
    }
    base[n] = (uint32_t)b;
    std::vector<float4> trav(n * PRT_INST_F4), shade(n * 8);
    auto                bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for(size_t j = 0; j < n; ++j)
    {
        const size_t        i  = order[j];
        const prt_instance& in = c->instances[i];
        const Tree&         t  = c->blas[in.blas];
        const bool          id = is_identity(in.transform);
        const prt_host::mat4 M = to_mat4(in.transform), W = prt_host::inverse(M);
        float4*              r = &trav[j * PRT_INST_F4];
        for(int k = 0; k < 3; ++k) r[k] = make_float4(W[0][k], W[1][k], W[2][k], W[3][k]);
        r[3] = make_float4(t.grid.qmin[0], t.grid.qmin[1], t.grid.qmin[2], t.sceneAbs);
        r[4] = make_float4(t.grid.qscale[0], t.grid.qscale[1], t.grid.qscale[2], bits((uint32_t)t.root));
        r[5] = make_float4(bits(base[i]), bits(id ? 1u : 0u), 0.0f, 0.0f);
        r[6] = r[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4* q = &shade[i * 8];
        for(int k = 0; k < 3; ++k)
        {
            q[k]     = make_float4(M[0][k], M[1][k], M[2][k], M[3][k]);
            q[3 + k] = make_float4(W[k][0], W[k][1], W[k][2], 0.0f);  // row k of transpose(inverse(M))
        }
        q[6] = make_float4(bits(t.first), bits(id ? 1u : 0u), 0.0f, 0.0f);
        q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // Which instance a flattened primitive id belongs to (k_shade): bucket table over the id range, about two buckets per
    // instance; entry k = the instance that holds id k << instShift, so the search for an id runs over
    // [table[id >> shift], table[(id >> shift) + 1]] — mostly one or two candidates instead of log2(n) steps.
    uint32_t shift = 0;
    while(((b >> shift) + 1) > 2 * (uint64_t)n + 2) ++shift;
    const size_t nb = (size_t)(b >> shift) + 2;
    base.resize(n + 1 + nb);
    for(size_t k = 0, i = 0; k < nb; ++k)
    {
        const uint64_t id = (uint64_t)k << shift;
        while(i + 1 < n && base[i + 1] <= id) ++i;
        base[n + 1 + k] = (uint32_t)i;
    }
    c->instShift = shift;
    // the tables stay where they are while they are large enough (an instance update of unchanged size moves nothing)
    if(trav.size() > c->instCap || !c->dInst || !c->dShadeInst)
    {
        dfree(c->dInst);
        dfree(c->dShadeInst);
        c->instCap = 0;
        HIPCHK(c, dalloc(c->dInst, trav.size()));
        HIPCHK(c, dalloc(c->dShadeInst, shade.size()));
        c->instCap = trav.size();
    }
    if(base.size() > c->instBaseCap || !c->dInstBase)
    {
        dfree(c->dInstBase);
        c->instBaseCap = 0;
        HIPCHK(c, dalloc(c->dInstBase, base.size()));
        c->instBaseCap = base.size();
    }
    HIPCHK(c, hipMemcpy(c->dInst, trav.data(), trav.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dShadeInst, shade.data(), shade.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dInstBase, base.data(), base.size() * 4, hipMemcpyHostToDevice));
    return PRT_OK;
}

// Node / leaf arrays of at least this size. They are kept when they are large enough, so that a refit or an instance
// update of unchanged size leaves the device addresses where they were (prt_trace_device_noalloc callers hold them in
// launches they enqueued themselves; the contents and the quantisation grid still change: see prt.h).
static hipError_t accel_reserve(prt_context* c, size_t nodes, size_t slots)
{
    hipError_t e = hipSuccess;
    if(nodes > c->nodesCap || !c->dNodes)
    {
        dfree(c->dNodes);
        c->nodesCap = 0;
        e = dalloc(c->dNodes, (nodes ? nodes : 1) * 4);
        if(e == hipSuccess) c->nodesCap = nodes ? nodes : 1;
    }
    if(e == hipSuccess && (slots > c->trisCap || !c->dTris))
    {
        dfree(c->dTris);
        c->trisCap = 0;
        e = dalloc(c->dTris, (size_t)PRT_LEAF_SLOT_F4 * (slots ? slots : 1));
        if(e == hipSuccess) c->trisCap = slots ? slots : 1;
    }
    return e;
}

// Traversal layout of all trees (after a build or a refit of the bottom level): for a two-level scene this (re)builds
// the top level from the current boxes of the bottom-level structures and the current transforms.
// topOnly (prt_update_instance_transforms, the top level's "updateOnly"): the bottom-level structures have not changed
// since the last emission, so their nodes and leaf records stay where they are and only the top level (tree, nodes,
// instance tables) is made again: O(instances), not O(scene). Node array of a two-level scene:
// [tops of the trees: top level | structures] [the structures' other nodes] [the top level's other nodes].
static int accel_emit(prt_context* c, bool topOnly = false)
{
    auto cleanup = [&]() {};
    int  rc;
    if(!c->twoLevel())
    {
        if((rc = tree_layout(c, c->top))) return rc;
        if(c->top.nInner >= (1u << 26)) return fail(c, PRT_E_LIMIT, "prt_build_accel: more than 2^26 BVH4 nodes");
        c->top.place = NodePlace{0, 0, 0, 0};
        c->nHot      = c->top.nInner < PRT_TOP_NODES ? c->top.nInner : PRT_TOP_NODES;
        TCHK(accel_reserve(c, c->top.nInner, c->top.N));
        tree_emit(c, c->top, true);
    }
    else
    {
        bool blasLaidOut = false;
        if(!topOnly)
        {
            // (root boxes of the structures: the top level is built from them)
            for(Tree& t : c->blas)
                if((rc = tree_layout(c, t))) return rc;
            blasLaidOut = true;
        }
        if((rc = tlas_build(c))) return rc;
        if((rc = tree_layout(c, c->top))) return rc;
        // hot parts first (LDS-staged by the traversal kernels): a third of the budget for the top level, the rest
        // shared by the bottom-level structures when they are few enough to get at least 4 nodes each
        const uint32_t budget = PRT_TOP_NODES;
        const uint32_t kTop   = c->top.nInner < budget / 3 ? c->top.nInner : budget / 3;
        const bool     keep   = topOnly && kTop == c->emittedTopHot && c->dNodes && c->dTris;
        if(!keep)
        {
            for(Tree& t : c->blas)
                if(!blasLaidOut && (rc = tree_layout(c, t))) return rc;
            const uint32_t kBlas = c->blas.size() * 4 <= budget - kTop ? (uint32_t)((budget - kTop) / c->blas.size()) : 0u;
            uint64_t       hot = kTop, sb = 0;
            for(Tree& t : c->blas)
            {
                const uint32_t k = t.nInner < kBlas ? t.nInner : kBlas;
                t.place          = NodePlace{(uint32_t)hot, k, 0, (uint32_t)sb};
                hot += k;
                sb += t.N;
            }
            uint64_t nb = hot;
            for(Tree& t : c->blas)
            {
                t.place.coldBase = (uint32_t)nb;
                nb += t.nInner - t.place.hotCount;
            }
            c->nHot        = (uint32_t)hot;
            c->topColdBase = (uint32_t)nb;
            // the top level's part is sized for the largest tree over this many instances (one leaf each: fewer inner
            // nodes than instances), so that moving the instances never moves the arrays
            if(nb + c->instances.size() + 1 >= (1u << 26) || sb >= (1u << 28))
                return fail(c, PRT_E_LIMIT, "prt_build_accel: more than 2^26 BVH4 nodes");
            TCHK(accel_reserve(c, (size_t)nb + c->instances.size() + 1, (size_t)sb));
            for(Tree& t : c->blas) tree_emit(c, t, true);
            c->emittedTopHot = kTop;
        }
        if(c->top.nInner - kTop > c->instances.size() + 1) return fail(c, PRT_E_LIMIT, "prt_build_accel: top level larger than its reserve");
        c->top.place = NodePlace{0, kTop, c->topColdBase, 0};
        tree_emit(c, c->top, false);
        TCHK(hipGetLastError());
        TCHK(hipStreamSynchronize(c->stream));
        if((rc = instance_tables(c))) return rc;
    }
    TCHK(hipGetLastError());
    TCHK(hipStreamSynchronize(c->stream));
    dfree(c->top.dWideIdx);
    for(Tree& t : c->blas) dfree(t.dWideIdx);
    return PRT_OK;
}
#undef TCHK

int prt_set_instances(prt_context* c, const prt_blas_range* blas, size_t nblas, const prt_instance* inst, size_t ninst)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveScene) return fail(c, PRT_E_NOSCENE, "prt_set_instances: no scene");
    if(ninst && (!blas || !inst || nblas == 0)) return fail(c, PRT_E_INVALID, "prt_set_instances: null input");
    if(ninst >= (1u << 24) || nblas >= (1u << 24)) return fail(c, PRT_E_LIMIT, "prt_set_instances: more than 2^24 instances");
    uint64_t total = 0;
    for(size_t k = 0; k < (ninst ? nblas : 0); ++k)
        if(blas[k].n_triangles == 0 || (uint64_t)blas[k].first_triangle + blas[k].n_triangles > c->ntri)
            return fail(c, PRT_E_INVALID, "prt_set_instances: range %zu (%u + %u) outside the scene's %u triangles", k,
                        blas[k].first_triangle, blas[k].n_triangles, c->ntri);
    for(size_t i = 0; i < ninst; ++i)
    {
        if(inst[i].blas >= nblas) return fail(c, PRT_E_INVALID, "prt_set_instances: instance %zu names structure %u of %zu", i, inst[i].blas, nblas);
        const float* m = inst[i].transform;
        if(m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f)
            return fail(c, PRT_E_INVALID, "prt_set_instances: instance %zu: the fourth row of the transform must be (0,0,0,1)", i);
        total += blas[inst[i].blas].n_triangles;
    }
    if(total >= 0xffffffffull) return fail(c, PRT_E_LIMIT, "prt_set_instances: more than 2^32 - 2 instanced triangles");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    free_accel(c);
    c->blasRanges.assign(blas, blas + (ninst ? nblas : 0));
    c->instances.assign(inst, inst + ninst);
    return PRT_OK;
}

int prt_build_accel(prt_context* c)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveScene) return fail(c, PRT_E_NOSCENE, "prt_build_accel: no scene");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const auto t0 = std::chrono::steady_clock::now();
    free_accel(c);
    int rc = PRT_OK;
    if(!c->twoLevel())
    {
        c->top.first = 0;
        c->top.N     = c->ntri;
        rc           = tree_build(c, c->top, nullptr, kLeafMax, PRT_MAX_DEPTH);
    }
    else
    {
        c->blas.resize(c->blasRanges.size());
        for(size_t k = 0; k < c->blas.size() && rc == PRT_OK; ++k)
        {
            c->blas[k].first = c->blasRanges[k].first_triangle;
            c->blas[k].N     = c->blasRanges[k].n_triangles;
            rc               = tree_build(c, c->blas[k], nullptr, kLeafMax, PRT_MAX_DEPTH);
        }
    }
    if(rc == PRT_OK) rc = accel_emit(c);
    if(rc)
    {
        const std::string msg = c->err;
        free_accel(c);
        c->err = msg;
        return rc;
    }
    c->haveAccel  = true;
    c->accelStale = false;
    c->buildMs    = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if(c->twoLevel() && c->gridTwo[0] == 0)
    {
        auto resident = [&](const void* fn) -> uint32_t {
            int nb = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, PRT_TRACE_BLOCK, 0) != hipSuccess || nb < 1) nb = 2;
            return (uint32_t)nb * (uint32_t)c->numCU;
        };
        c->gridTwo[0] = resident((const void*)k_extend<false, true>);
        c->gridTwo[1] = resident((const void*)k_shadow<false, true>);
        c->gridTwo[2] = resident((const void*)k_trace_batch<false, true>);
        c->gridTwo[3] = resident((const void*)k_ao_shade<true>);
        c->gridTwo[4] = resident((const void*)k_shadow_extend<false, true>);
        {
            int nb = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_shade<true>, PRT_SHADE_BLOCK, 0) != hipSuccess || nb < 1) nb = 2;
            c->gridTwo[5] = (uint32_t)nb * (uint32_t)c->numCU;
        }
    }
    return PRT_OK;
}

// prt_update_vertices + prt_refit_accel: the "updateOnly" path of the reference's AS helpers
// (vkRT_BLAS.cpp:209-290, vkRT_TLAS.cpp:178-279 take updateOnly; vkRTX_setup.cpp never sets it)
int prt_update_vertices(prt_context* c, const prt_vertex* v, size_t nv)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveScene) return fail(c, PRT_E_NOSCENE, "prt_update_vertices: no scene");
    if(!v || nv != c->nverts)
        return fail(c, PRT_E_INVALID, "prt_update_vertices: %zu vertices given, the scene has %u", nv, c->nverts);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemcpy(c->dVerts, v, nv * sizeof(prt_vertex), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather_shading, dim3(cdiv(c->ntri, 256)), dim3(256), 0, c->stream, c->dVerts, c->dIdx, c->ntri,
                       c->dShadeTris);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->accelStale = c->haveAccel;  // rendering refuses to run until prt_refit_accel or prt_build_accel
    return PRT_OK;
}

// boxes of one tree over triangles from the current vertices: leaves first, then level by level from the deepest one
static void tree_refit(prt_context* c, Tree& t)
{
    hipLaunchKernelGGL(k_refit_leaves, dim3(cdiv(t.nNodes, 256)), dim3(256), 0, c->stream, t.dBuildNodes, t.nNodes,
                       c->dVerts, c->dIdx + (size_t)3 * t.first, t.dOrder);
    for(uint32_t l = t.lt.n; l-- > 0;)
    {
        const uint32_t start = t.lt.start[l];
        const uint32_t end   = l + 1 < t.lt.n ? t.lt.start[l + 1] : t.nNodes;
        if(end > start)
            hipLaunchKernelGGL(k_refit_level, dim3(cdiv(end - start, 256)), dim3(256), 0, c->stream, t.dBuildNodes, start,
                               end - start);
    }
}

int prt_refit_accel(prt_context* c)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_refit_accel: no acceleration structure to update");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const auto t0 = std::chrono::steady_clock::now();
    if(!c->twoLevel())
        tree_refit(c, c->top);
    else
        for(Tree& t : c->blas) tree_refit(c, t);  // the top level is rebuilt from the new boxes by accel_emit
    HIPCHK(c, hipGetLastError());
    const int rc = accel_emit(c);
    if(rc)
    {
        c->haveAccel = false;  // the arrays may be partly rewritten or gone: nothing may be traced against them
        return rc;
    }
    c->accelStale = false;
    c->refitMs    = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PRT_OK;
}

// The "updateOnly" path of the top level (vkRT_TLAS.cpp:178-279 with updateOnly = true): new transforms for the
// instances of prt_set_instances (n_instances x 16 floats, column-major); the bottom-level structures stay.
int prt_update_instance_transforms(prt_context* c, const float* transforms, size_t ninst)
{
    if(!c) return PRT_E_INVALID;
    if(!c->twoLevel()) return fail(c, PRT_E_INVALID, "prt_update_instance_transforms: the scene has no instances");
    if(!transforms || ninst != c->instances.size())
        return fail(c, PRT_E_INVALID, "prt_update_instance_transforms: %zu transforms given, the scene has %zu instances",
                    ninst, c->instances.size());
    for(size_t i = 0; i < ninst; ++i)
    {
        const float* m = transforms + 16 * i;
        if(m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f)
            return fail(c, PRT_E_INVALID, "prt_update_instance_transforms: instance %zu: the fourth row of the transform must be (0,0,0,1)", i);
    }
    // nothing is committed before every transform has been checked: a refused update leaves the scene as it was
    for(size_t i = 0; i < ninst; ++i)
    {
        double kappa;
        if(!instance_condition(transforms + 16 * i, kappa))
            return fail(c, PRT_E_INVALID, "prt_update_instance_transforms: the transform of instance %zu is singular or too ill-conditioned (condition %g)", i, kappa);
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    for(size_t i = 0; i < ninst; ++i) memcpy(c->instances[i].transform, transforms + 16 * i, 64);
    if(!c->haveAccel) return PRT_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const int  rc = accel_emit(c, !c->accelStale);  // top level only, unless moved vertices await their refit
    if(rc)
    {
        c->haveAccel = false;
        return rc;
    }
    c->refitMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PRT_OK;
}

int prt_set_scramble_seed(prt_context* c, uint32_t seed)
{
    if(!c) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    dfree(c->dScramble);
    c->seed = seed;
    return PRT_OK;
}

int prt_upload_scrambles(prt_context* c, const uint32_t* layers, uint32_t nlayers)
{
    if(!c) return PRT_E_INVALID;
    if(!layers || nlayers != 32) return fail(c, PRT_E_INVALID, "prt_upload_scrambles: need 32 layers");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    dfree(c->dScramble);
    HIPCHK(c, dalloc(c->dScramble, (size_t)32 * c->npix));
    HIPCHK(c, hipMemcpy(c->dScramble, layers, (size_t)32 * c->npix * 4, hipMemcpyHostToDevice));
    return PRT_OK;
}

int prt_set_uniforms(prt_context* c, const prt_uniforms* u)
{
    if(!c || !u) return PRT_E_INVALID;
    if(u->numIndirectBounces < 0 || u->numIndirectBounces > (int)kMaxBounces)
        return fail(c, PRT_E_LIMIT, "prt_set_uniforms: numIndirectBounces %d outside [0,%u]", u->numIndirectBounces,
                    kMaxBounces);
    if(u->numAArays < 0 || u->numAArays > 255) return fail(c, PRT_E_LIMIT, "prt_set_uniforms: numAArays %d", u->numAArays);
    if(u->numAOrays < 0) return fail(c, PRT_E_INVALID, "prt_set_uniforms: numAOrays %d", u->numAOrays);
    c->ubo = *u;
    return PRT_OK;
}

int prt_set_flags(prt_context* c, uint32_t flags)
{
    if(!c) return PRT_E_INVALID;
    c->flags = flags & (PRT_FLAG_LITERAL | PRT_FLAG_RUSSIAN_ROULETTE);
    return PRT_OK;
}

static hipEvent_t prof_event(prt_context* c)
{
    if(!c->eventPool.empty())
    {
        hipEvent_t e = c->eventPool.back();
        c->eventPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_collect(prt_context* c)
{
    if(c->pending.empty()) return;
    (void)sync_all(c);
    for(auto& p : c->pending)
    {
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess)
        {
            c->profMs[p.kind] += ms;
            c->profLaunches[p.kind]++;
        }
        c->eventPool.push_back(p.e0);
        c->eventPool.push_back(p.e1);
    }
    c->pending.clear();
}
// brackets one launch with events when profiling is on
struct ProfScope
{
    prt_context* c;
    int          kind;
    hipStream_t  st;
    hipEvent_t   e0 = nullptr;
    ProfScope(prt_context* c_, int k, hipStream_t s_) : c(c_), kind(k), st(s_)
    {
        if(c->profLevel > 0 && ((c->profMask >> k) & 1u))
        {
            e0 = prof_event(c);
            (void)hipEventRecord(e0, st);
        }
    }
    ~ProfScope()
    {
        if(e0)
        {
            hipEvent_t e1 = prof_event(c);
            (void)hipEventRecord(e1, st);
            c->pending.push_back({kind, e0, e1});
        }
    }
};

static void fill_bvh_view(const prt_context* c, BvhView& b)
{
    b.nodes    = c->dNodes;
    b.tris     = c->dTris;
    b.root     = c->top.root;
    b.sceneAbs = c->top.sceneAbs;
    for(int k = 0; k < 3; ++k)
    {
        b.qmin[k]   = c->top.grid.qmin[k];
        b.qscale[k] = c->top.grid.qscale[k];
    }
    b.inst     = c->dInst;
    b.padScale = c->padScale;
    {
        uint64_t inner = c->nHot;  // the node array starts with the tops of the trees (accel_emit)
        static const uint32_t topMax = [] {
            uint32_t v = PRT_TOP_NODES;
            if(const char* e = getenv("PRT_TOP")) v = (uint32_t)atoi(e) < v ? (uint32_t)atoi(e) : v;  // measurement only
            return v;
        }();
        b.nTop = (uint32_t)(inner < topMax ? inner : topMax);
    }
}

static void make_views(prt_context* c, const Pipe& p, SceneView& sc, FrameView& f, uint32_t nsub = 1, uint32_t iterStride = 1)
{
    sc.stri         = c->dShadeTris;
    sc.sobolBytes   = c->dSobolBytes;
    sc.verts        = c->dVerts;
    sc.idx          = c->dIdx;
    sc.mats         = c->dMats;
    sc.tex          = c->dTex;
    sc.sobol        = c->dSobol;
    sc.scrTable     = c->dScramble;
    sc.nmats        = c->nmats;
    sc.ntex         = c->ntex;
    sc.seed         = c->seed;
    sc.ntri         = c->ntri;
    sc.shadeInst    = c->dShadeInst;
    sc.instBase     = c->dInstBase;
    sc.instLookup   = c->dInstBase ? c->dInstBase + c->instances.size() + 1 : nullptr;
    sc.instShift    = c->instShift;
    sc.ninst        = (uint32_t)c->instances.size();
    fill_bvh_view(c, sc.bvh);
    f.w       = c->w;
    f.h       = c->h;
    f.npix    = c->npix * nsub;
    f.npixImg = c->npix;
    f.iterStride = iterStride;
    f.accum   = c->accum;
    for(int g = 0; g < 2; ++g) f.gen[g] = PathGen{p.rayO[g], p.rayD[g], p.hit[g], p.thr[g], p.q[g]};
    f.shO     = p.shO;
    f.shD     = p.shD;
    f.eiVis   = p.eiVis;
    f.shPx    = p.qShadow;
    f.contrib = p.contrib;
    f.aaOff   = p.aaOff;
    f.hitIds  = p.hitIds;
    f.dead    = p.dead;
    f.nvert   = p.nvert;
    f.stats   = c->stats;
}

static void launch_extend(prt_context* c, hipStream_t st, const SceneView& sc, const FrameView& f, uint32_t grid,
                          uint32_t q /* generation of path state */, const uint32_t* n, uint32_t* work, uint32_t primary)
{
    ProfScope ps(c, PRT_K_EXTEND, st);
    hipLaunchKernelGGL(kExtendTable.pick(c), dim3(c->twoLevel() ? c->gridTwo[0] : grid), dim3(PRT_TRACE_BLOCK), 0, st, sc, f, q,
                       n, work, primary, trace_tune());
}

static int pipe_ensure(prt_context* c, Pipe& p, size_t npass, int B);
// AO.rgen: generate -> extend -> AO rays + accumulate, on the context's main stream with pipe 0's buffers.
// nsub > 1: that many iterations side by side in one pass (prt_render; not with the literal image semantics): every
// iteration's term is stored by pass pixel and k_resolve adds them to the image in iteration order.
static int render_ao(prt_context* c, uint32_t nsub = 1)
{
    Pipe&       p  = c->pipes[0];
    hipStream_t st = c->stream;
    {
        // (also for one iteration: a pass that could not be allocated leaves pipe 0 without buffers)
        const int rc = pipe_ensure(c, p, (size_t)c->npix * nsub, 0);
        if(rc) return rc;
    }
    SceneView   sc;
    FrameView   f;
    make_views(c, p, sc, f, nsub, 1);
    const uint32_t gridPix = cdiv((size_t)c->npix * nsub, 256);
    HIPCHK(c, hipMemsetAsync(p.counters, 0, kCounterWords * 4, st));
    {
        ProfScope ps(c, PRT_K_GENERATE, st);
        hipLaunchKernelGGL(k_ao_generate, dim3(gridPix), dim3(256), 0, st, f, c->ubo, c->flags);
    }
    launch_extend(c, st, sc, f, c->gridExtend, 0u, nullptr, p.counters + 0, 1u);
    {
        ProfScope ps(c, PRT_K_AO_SHADE, st);
        // k_ao_shade<TWO, BATCH>
        static const decltype(&k_ao_shade<false, false>) kAo[2][2] = {{k_ao_shade<false, false>, k_ao_shade<false, true>},
                                                                      {k_ao_shade<true, false>, k_ao_shade<true, true>}};
        const bool two = c->twoLevel();
        hipLaunchKernelGGL(kAo[two ? 1 : 0][nsub > 1 ? 1 : 0], dim3(two ? c->gridTwo[3] : c->gridAo), dim3(PRT_TRACE_BLOCK), 0, st,
                           sc, f, c->ubo, c->flags, p.counters + 1);
    }
    if(nsub > 1)
    {
        ProfScope ps(c, PRT_K_RESOLVE, st);
        hipLaunchKernelGGL(k_resolve, dim3(cdiv(c->npix, 256)), dim3(256), 0, st, c->accum, p.contrib, p.nvert, c->npix, nsub);
    }
    HIPCHK(c, hipGetLastError());
    c->lastPipe = 0;
    p.lastSub   = nsub;  // prt_read_hit_ids: the ids of the pass's last iteration
    return PRT_OK;
}

// One pass of pathRT.rgen (one aaRay of one vkCmdTraceRaysNV(W,H,1)): a fixed launch sequence on the pipe's
// stream, then k_resolve on the main stream (iteration order). No host synchronisation.
// buffers of one pipe for a pass over npass pixels with B bounces
// Device bytes of a pipe's buffers per pixel of a pass with B bounces (pipe_alloc + the contrib planes)
static size_t pass_bytes_per_pixel(int B)
{
    return 2 * 4 * sizeof(float4)      // rayO, rayD, hit, thr: two generations
           + 3 * sizeof(float4)        // shO, shD, eiVis
           + sizeof(float2) + 4 + 1 + 1  // aaOff, hitIds, dead, nvert
           + 3 * 4                     // q[0], q[1], qShadow
           + (size_t)(B + 1) * sizeof(float4);  // contrib[bounce]
}
static int pipe_ensure(prt_context* c, Pipe& p, size_t npass, int B)
{
    if(p.allocated && p.cap >= npass && p.contribCap >= (uint32_t)B + 1u) return PRT_OK;
    if(p.allocated) HIPCHK(c, sync_all(c));
    c->oom = false;
    if(c->passBytesMax && npass > c->npix && (uint64_t)npass * pass_bytes_per_pixel(B) > c->passBytesMax)
    {
        c->oom = true;  // over the caller's bound: treated like an allocation that failed (prt_render shrinks the pass)
        return fail(c, PRT_E_HIP, "pass of %zu pixels needs %zu bytes, limit %llu", npass, npass * pass_bytes_per_pixel(B),
                    (unsigned long long)c->passBytesMax);
    }
    const bool grow = p.cap < npass;
    hipError_t e    = pipe_alloc(p, npass);
    if(e == hipSuccess && (grow || p.contribCap < (uint32_t)B + 1u))
    {
        dfree(p.contrib);
        p.contribCap = 0;
        e            = dalloc(p.contrib, (size_t)(B + 1) * p.cap);
        if(e == hipSuccess) p.contribCap = (uint32_t)B + 1u;
    }
    if(e != hipSuccess)
    {
        // what was allocated goes back, so that a smaller pass can be tried; the runtime's sticky error is taken off
        pipe_free_buffers(p);
        c->oom = e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation;
        (void)hipGetLastError();
        return fail(c, PRT_E_HIP, "pipe allocation for %zu pixels failed: %s", npass, hipGetErrorString(e));
    }
    return PRT_OK;
}

static int render_path_pass(prt_context* c, int pi, int aa, uint32_t nsub = 1, uint32_t iterStride = 1, bool rotate = true)
{
    Pipe&        p     = c->pipes[pi];
    const size_t npass = (size_t)c->npix * nsub;  // "pixels" of this pass: nsub iterations side by side
    const int    B     = c->ubo.numIndirectBounces;
    // all pipes of the rotation at once: an allocation synchronises the device, so it must not happen in the middle
    // of a render (the first pass on pipe 0 would otherwise run alone, then wait for pipe 1's buffers, ...)
    for(int q = 0; rotate && q < c->nPipes; ++q)
    {
        const int rc = pipe_ensure(c, c->pipes[q], npass, B);
        if(rc) return rc;
    }
    if(!rotate || pi >= c->nPipes)
    {
        const int rc = pipe_ensure(c, p, npass, B);
        if(rc) return rc;
    }
    hipStream_t st = p.stream;
    SceneView   sc;
    FrameView   f;
    make_views(c, p, sc, f, nsub, iterStride);
    if(p.busy) HIPCHK(c, hipStreamWaitEvent(st, p.evFree, 0));  // the terms of its previous pass have been summed
    const uint32_t gridPix   = cdiv(npass, 256);
    const uint32_t gridShade = c->gridShade;
    // counters: [0..31] path-queue lengths per bounce, [32..63] shadow-queue lengths,
    //           [64..127] work-fetch cursors of the extend launches, [128..191] of the shadow launches
    HIPCHK(c, hipMemsetAsync(p.counters, 0, kCounterWords * 4, st));
    const uint32_t fl = c->flags | ((uint32_t)aa << 8);
    {
        ProfScope ps(c, PRT_K_GENERATE, st);
        hipLaunchKernelGGL(k_generate, dim3(gridPix), dim3(256), 0, st, sc, f, c->ubo, c->flags, (uint32_t)aa);
    }
    for(int b = 0; b <= B; ++b)
    {
        const uint32_t* nIn = b == 0 ? nullptr : p.counters + b;  // vertex b's paths: slots 0 .. *nIn-1 of generation b & 1
        // with fusion the extension rays of bounce b >= 1 were traced together with the shadow rays of bounce b-1
        if(b == 0 || !c->fuse) launch_extend(c, st, sc, f, c->gridExtend, (uint32_t)b, nIn, p.counters + 64 + b, b == 0 ? 1u : 0u);
        {
            ProfScope ps(c, PRT_K_SHADE, st);
            const bool two = c->twoLevel();
            hipLaunchKernelGGL(two ? k_shade<true> : k_shade<false>, dim3(two ? c->gridTwo[5] : gridShade), dim3(PRT_SHADE_BLOCK), 0, st, sc, f,
                               c->ubo, fl, (uint32_t)b, nIn, p.counters + b + 1, p.counters + 32 + b);
        }
        if(c->fuse && b < B)
        {
            ProfScope ps(c, PRT_K_SHADOW_EXTEND, st);
            hipLaunchKernelGGL(kShadowExtendTable.pick(c), dim3(c->twoLevel() ? c->gridTwo[4] : c->gridFused), dim3(PRT_TRACE_BLOCK),
                               0, st, sc, f, (const uint32_t*)(p.counters + 32 + b), p.counters + 128 + b, (uint32_t)b,
                               (const uint32_t*)(p.counters + b + 1), p.counters + 64 + b + 1, trace_tune());
        }
        else
        {
            ProfScope ps(c, PRT_K_SHADOW, st);
            hipLaunchKernelGGL(kShadowTable.pick(c), dim3(c->twoLevel() ? c->gridTwo[1] : c->gridShadow), dim3(PRT_TRACE_BLOCK), 0,
                               st, sc, f, (const uint32_t*)(p.counters + 32 + b), p.counters + 128 + b, (uint32_t)b, trace_tune());
        }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(p.evDone, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream, p.evDone, 0));
    {
        ProfScope ps(c, PRT_K_RESOLVE, c->stream);
        hipLaunchKernelGGL(k_resolve, dim3(cdiv(c->npix, 256)), dim3(256), 0, c->stream, c->accum,
                           (const float4*)p.contrib, (const uint8_t*)p.nvert, c->npix, nsub);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(p.evFree, c->stream));
    p.busy      = true;
    p.lastSub   = nsub;
    c->lastPipe = pi;
    if(c->pending.size() > 8192) prof_collect(c);
    return PRT_OK;
}

int prt_render_iteration(prt_context* c, int mode)
{
    if(!c) return PRT_E_INVALID;
    if(mode != PRT_MODE_PATH && mode != PRT_MODE_AO) return fail(c, PRT_E_INVALID, "prt_render_iteration: mode %d", mode);
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_render_iteration: no acceleration structure");
    if(c->accelStale)
        return fail(c, PRT_E_NOSCENE, "prt_render_iteration: vertices changed, call prt_refit_accel or prt_build_accel");
    HIPCHK(c, hipSetDevice(c->device));
    if(mode != c->lastMode)
    {
        HIPCHK(c, sync_all(c));  // AO works on pipe 0's buffers from the main stream
        c->lastMode = mode;
    }
    int rc = PRT_OK;
    if(mode == PRT_MODE_AO)
        rc = render_ao(c);
    else
    {
        // Passes are independent up to k_resolve, so consecutive iterations run on different pipes. The aaRays of
        // one iteration depend on each other through `dead` (a primary miss ends the pixel's iteration), and the
        // literal image semantics are defined pass after pass: both stay on pipe 0.
        const bool pipelined = c->nPipes > 1 && c->ubo.numAArays == 1 && !(c->flags & PRT_FLAG_LITERAL);
        if(pipelined)
        {
            rc          = render_path_pass(c, c->nextPipe, 0);
            c->nextPipe = (c->nextPipe + 1) % c->nPipes;
        }
        else
            for(int aa = 0; aa < c->ubo.numAArays && rc == PRT_OK; ++aa) rc = render_path_pass(c, 0, aa);
    }
    if(rc == PRT_OK) c->iterations++;
    return rc;
}

int prt_render(prt_context* c, int mode, uint32_t first, uint32_t n, uint32_t stride)
{
    if(!c) return PRT_E_INVALID;
    if(stride == 0) return fail(c, PRT_E_INVALID, "prt_render: stride 0");
    // Several iterations per pass: a launch over B*W*H "pixels" has the same fixed tail as one over W*H, so small
    // frames batch consecutive iterations (pixel k of the pass = image pixel k % (W*H) of iteration
    // first + (k / (W*H)) * stride); k_resolve adds them in iteration order, so the image is bit-identical.
    const bool batchable = (mode == PRT_MODE_PATH || mode == PRT_MODE_AO) && c->batch > 1 &&
                           (mode == PRT_MODE_AO || c->ubo.numAArays == 1) && !(c->flags & PRT_FLAG_LITERAL) && c->haveAccel &&
                           !c->accelStale;
    uint32_t k = 0;
    if(batchable)
    {
        HIPCHK(c, hipSetDevice(c->device));
        if(mode != c->lastMode)
        {
            HIPCHK(c, sync_all(c));
            c->lastMode = mode;
        }
        // Memory policy: a pass that cannot be allocated (hipErrorOutOfMemory, or more than prt_set_pass_memory_limit /
        // PRT_MAX_PASS_BYTES allows) is retried with half the iterations, down to one iteration per pass — the one-by-one
        // path below, whose buffers exist since prt_create. The size that fitted is remembered (batchFit).
        uint32_t cap = c->batchFit ? (c->batchFit < c->batch ? c->batchFit : c->batch) : c->batch;
        while(n - k >= 2 && cap >= 2)
        {
            uint32_t nsub = n - k < cap ? n - k : cap;
            if((uint64_t)c->npix * nsub >= (1ull << 31)) nsub = (uint32_t)(((1ull << 31) - 1) / c->npix);  // 32-bit pixel ids
            if(nsub < 2) break;
            c->ubo.iteration    = first + k * stride;
            int rc;
            if(mode == PRT_MODE_AO)  // (AO.rgen takes nothing from the iteration number: the iterations of a pass are alike)
                rc = render_ao(c, nsub);
            else
            {
                // A large pass fills the GPU by itself: passes in flight on other streams add nothing then (measured at
                // 1080p, 32 / 64 iterations per pass: 8170 / 8280 Mrays/s on one pipe, 8120 / 8250 on three), so such
                // passes all run on pipe 0 and only that pipe's buffers grow to the pass size.
                const bool large = (uint64_t)c->npix * nsub >= (24ull << 20);
                rc               = render_path_pass(c, large ? 0 : c->nextPipe, 0, nsub, stride, !large);
                if(rc == PRT_OK && !large) c->nextPipe = (c->nextPipe + 1) % c->nPipes;
            }
            if(rc != PRT_OK)
            {
                if(!c->oom) return rc;
                c->oom      = false;
                cap         = nsub / 2;
                c->batchFit = cap < 1 ? 1 : cap;
                continue;
            }
            c->batchUsed = nsub;
            c->iterations += nsub;
            k += nsub;
        }
        if(k < n) c->batchUsed = c->batchUsed ? c->batchUsed : 1;
    }
    for(; k < n; ++k)
    {
        c->ubo.iteration = first + k * stride;
        const int rc     = prt_render_iteration(c, mode);
        if(rc) return rc;
    }
    return PRT_OK;
}

int prt_set_iteration_batch(prt_context* c, int n)
{
    if(!c) return PRT_E_INVALID;
    if(n < 1 || n > 64) return fail(c, PRT_E_LIMIT, "prt_set_iteration_batch: %d outside [1,64]", n);
    c->batch    = (uint32_t)n;
    c->batchFit = 0;
    return PRT_OK;
}
int prt_get_iteration_batch(const prt_context* c) { return c ? (int)c->batch : PRT_E_INVALID; }
int prt_get_pass_iterations(const prt_context* c) { return c ? (int)c->batchUsed : PRT_E_INVALID; }
int prt_set_pass_memory_limit(prt_context* c, uint64_t bytes)
{
    if(!c) return PRT_E_INVALID;
    c->passBytesMax = bytes;
    c->batchFit     = 0;
    return PRT_OK;
}

int prt_get_pipeline_depth(const prt_context* c) { return c ? c->nPipes : PRT_E_INVALID; }

int prt_set_pipeline_depth(prt_context* c, int n)
{
    if(!c) return PRT_E_INVALID;
    if(n < 1 || n > kMaxPipes) return fail(c, PRT_E_LIMIT, "prt_set_pipeline_depth: %d outside [1,%d]", n, kMaxPipes);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    c->nPipes   = n;
    c->nextPipe = 0;
    return PRT_OK;
}

int prt_sync(prt_context* c)
{
    if(!c) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    return PRT_OK;
}

int prt_clear_accum(prt_context* c)
{
    if(!c) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->accum, 0, (size_t)c->npix * sizeof(float4), c->stream));
    return PRT_OK;
}

int prt_read_accum(prt_context* c, float* out)
{
    if(!c || !out) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->accum, (size_t)c->npix * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_write_accum(prt_context* c, const float* in)
{
    if(!c || !in) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemcpyAsync(c->accum, in, (size_t)c->npix * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_device_accum_ptr(prt_context* c, void** p)
{
    if(!c || !p) return PRT_E_INVALID;
    *p = c->accum;
    return PRT_OK;
}

int prt_bind_accum(prt_context* c, void* p)
{
    if(!c) return PRT_E_INVALID;
    if(((uintptr_t)p & 15u) != 0) return fail(c, PRT_E_INVALID, "prt_bind_accum: pointer not 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    c->accum = p ? (float4*)p : c->accumOwn;
    return PRT_OK;
}

int prt_postprocess(prt_context* c, uint8_t* out)
{
    if(!c || !out) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t* d = nullptr;
    HIPCHK(c, dalloc(d, (size_t)c->npix));
    hipLaunchKernelGGL(k_postprocess, dim3(cdiv(c->npix, 256)), dim3(256), 0, c->stream, (const float4*)c->accum,
                       c->npix, d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)c->npix * 4, hipMemcpyDeviceToHost, c->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dfree(d);
    HIPCHK(c, e);
    return PRT_OK;
}

int prt_read_hit_ids(prt_context* c, uint32_t* out)
{
    if(!c || !out) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const Pipe& lp = c->pipes[c->lastPipe];  // the last iteration of the pipe's last pass
    HIPCHK(c, hipMemcpy(out, lp.hitIds + (size_t)(lp.lastSub - 1) * c->npix, (size_t)c->npix * 4, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_get_stats(prt_context* c, prt_stats* s)
{
    if(!c || !s) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long rays[2] = {0, 0}, overruns = 0;
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemcpy(rays, c->stats, 16, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&overruns, c->stats + 28, 8, hipMemcpyDeviceToHost));
    s->rays            = rays[0] + rays[1];
    s->iterations      = c->iterations;
    s->bvh_nodes       = c->top.nNodes;
    s->bvh_inner_nodes = c->top.nInner;
    s->bvh_depth       = c->top.depth;
    {
        uint32_t deepest = 0;
        for(const Tree& t : c->blas)  // two-level scenes: all structures; depth = top level + deepest bottom level
        {
            s->bvh_nodes += t.nNodes;
            s->bvh_inner_nodes += t.nInner;
            deepest = t.depth > deepest ? t.depth : deepest;
        }
        s->bvh_depth += deepest;
    }
    s->bvh_build_ms    = c->buildMs;
    s->bvh_refit_ms    = c->refitMs;
    s->trace_overruns  = (uint32_t)(overruns > 0xffffffffull ? 0xffffffffull : overruns);
    return PRT_OK;
}

int prt_reset_stats(prt_context* c)
{
    if(!c) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    prof_collect(c);
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemsetAsync(c->stats, 0, kStatWords * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->iterations = 0;
    for(int k = 0; k < PRT_K_COUNT; ++k)
    {
        c->profMs[k]       = 0.0;
        c->profLaunches[k] = 0;
    }
    return PRT_OK;
}

int prt_set_profiling(prt_context* c, int level)
{
    if(!c || level < 0 || level > 2) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    prof_collect(c);
    c->profLevel = level;
    return PRT_OK;
}

int prt_set_profiling_mask(prt_context* c, uint32_t kinds)
{
    if(!c) return PRT_E_INVALID;
    c->profMask = kinds;
    return PRT_OK;
}

int prt_get_profile(prt_context* c, prt_profile* out)
{
    if(!c || !out) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    prof_collect(c);
    unsigned long long st[kStatWords];
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemcpy(st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    for(int k = 0; k < PRT_K_COUNT; ++k)
    {
        out->ms[k]       = c->profMs[k];
        out->launches[k] = c->profLaunches[k];
    }
    out->closest_rays  = st[0];
    out->any_rays      = st[1];
    out->closest_nodes = st[2];
    out->closest_tris  = st[3];
    out->any_nodes     = st[4];
    out->any_tris      = st[5];
    out->fused_closest_rays = st[6];
    out->fused_any_rays     = st[7];
    for(int k = 0; k < 2; ++k)
    {
        prt_wave_steps& w = k == 0 ? out->closest_steps : out->any_steps;
        w.node_steps      = st[8 + 6 * k];
        w.node_lanes      = st[9 + 6 * k];
        w.tri_steps       = st[10 + 6 * k];
        w.tri_lanes       = st[11 + 6 * k];
        w.setup_steps     = st[12 + 6 * k];
        w.setup_lanes     = st[13 + 6 * k];
        w.setup_ticks     = st[20 + 4 * k];
        w.node_ticks      = st[21 + 4 * k];
        w.tri_ticks       = st[22 + 4 * k];
        w.total_ticks     = st[23 + 4 * k];
    }
    return PRT_OK;
}

static void launch_trace_batch(prt_context* c, hipStream_t st, const BvhView& bvh, const void* rays, size_t n, int any,
                               void* hits, uint32_t* work)
{
    // k_trace_batch<ANY, TWO>
    static const decltype(&k_trace_batch<false, false>) kBatch[2][2] = {{k_trace_batch<false, false>, k_trace_batch<false, true>},
                                                                        {k_trace_batch<true, false>, k_trace_batch<true, true>}};
    const bool two = c->twoLevel();
    hipLaunchKernelGGL(kBatch[any ? 1 : 0][two ? 1 : 0], dim3(two ? c->gridTwo[2] : c->gridBatch), dim3(PRT_TRACE_BLOCK), 0, st, bvh,
                       (const float4*)rays, (uint32_t)n, (float4*)hits, work, trace_tune(), c->stats);
}

int prt_trace_device(prt_context* c, const void* rays, size_t n, int any, void* hits, float* ms)
{
    if(!c || !rays || !hits) return PRT_E_INVALID;
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_trace: no acceleration structure");
    if(c->accelStale) return fail(c, PRT_E_NOSCENE, "prt_trace: vertices changed, call prt_refit_accel or prt_build_accel");
    if(n >= (1ull << 31)) return fail(c, PRT_E_LIMIT, "prt_trace: too many rays");
    HIPCHK(c, hipSetDevice(c->device));
    if(n == 0) { if(ms) *ms = 0.0f; return PRT_OK; }
    SceneView sc;
    FrameView f;
    make_views(c, c->pipes[0], sc, f);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    uint32_t*      work = c->batchCounter;  // work-fetch cursor of this launch
    HIPCHK(c, hipMemsetAsync(work, 0, 4, c->stream));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    launch_trace_batch(c, c->stream, sc.bvh, rays, n, any, hits, work);
    hipError_t e = hipGetLastError();
    if(e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if(e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0.0f;
    if(e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(c, e);
    if(ms) *ms = t;
    return PRT_OK;
}

// prt_trace_device without any allocation: no hipMalloc / hipFree / event or stream creation; the kernel's work-fetch
// cursor is this GPU's copy of the file-scope `scratch` array. Asynchronous on `stream` (NULL: the context's).
int prt_trace_device_noalloc(prt_context* c, const void* rays, size_t n, int any, void* hits, void* stream)
{
    if(!c || !rays || !hits) return PRT_E_INVALID;
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_trace_device_noalloc: no acceleration structure");
    if(c->accelStale)
        return fail(c, PRT_E_NOSCENE, "prt_trace_device_noalloc: vertices changed, call prt_refit_accel or prt_build_accel");
    if(n > PRT_NOALLOC_MAX_RAYS)
        return fail(c, PRT_E_LIMIT, "prt_trace_device_noalloc: %zu rays, the limit of this entry point is %u", n,
                    (unsigned)PRT_NOALLOC_MAX_RAYS);
    if(c->device < 0 || c->device >= kMaxDevices || !g_scratchPtr[c->device])
        return fail(c, PRT_E_HIP, "prt_trace_device_noalloc: scratch of device %d was not set up", c->device);
    HIPCHK(c, hipSetDevice(c->device));
    if(n == 0) return PRT_OK;
    uint32_t*   work = g_scratchPtr[c->device];  // never the array's name: that is a host-side stand-in
    hipStream_t st   = stream ? (hipStream_t)stream : c->stream;
    BvhView     bvh;
    fill_bvh_view(c, bvh);
    // the array keeps the last call's cursor: it must start at zero
    HIPCHK(c, hipMemsetAsync(work, 0, PRT_NOALLOC_SCRATCH_BYTES, st));
    launch_trace_batch(c, st, bvh, rays, n, any, hits, work);
    HIPCHK(c, hipGetLastError());
    return PRT_OK;
}

int prt_trace(prt_context* c, const prt_ray* rays, size_t n, int any, prt_hit* out)
{
    if(!c || (n && (!rays || !out))) return PRT_E_INVALID;
    if(n == 0) return PRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float4 *dr = nullptr, *dh = nullptr;
    HIPCHK(c, dalloc(dr, 2 * n));
    hipError_t e = dalloc(dh, n);
    if(e == hipSuccess) e = hipMemcpy(dr, rays, n * sizeof(prt_ray), hipMemcpyHostToDevice);
    int rc = PRT_OK;
    if(e == hipSuccess) rc = prt_trace_device(c, dr, n, any, dh, nullptr);
    if(e == hipSuccess && rc == PRT_OK) e = hipMemcpy(out, dh, n * sizeof(prt_hit), hipMemcpyDeviceToHost);
    dfree(dr);
    dfree(dh);
    if(rc) return rc;
    HIPCHK(c, e);
    return PRT_OK;
}

static int read_tree(prt_context* c, const Tree& t, prt_bvh_node* nodes, uint32_t* order)
{
    HIPCHK(c, hipSetDevice(c->device));
    if(nodes)
    {
        float4* d = nullptr;
        HIPCHK(c, dalloc(d, (size_t)2 * t.nNodes));
        hipLaunchKernelGGL(k_decode_nodes, dim3(cdiv(t.nNodes, 256)), dim3(256), 0, c->stream, t.dBuildNodes, t.nNodes, d);
        hipError_t e = hipMemcpyAsync(nodes, d, (size_t)t.nNodes * 32, hipMemcpyDeviceToHost, c->stream);
        if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
        dfree(d);
        HIPCHK(c, e);
    }
    if(order) HIPCHK(c, hipMemcpy(order, t.dOrder, (size_t)t.N * 4, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_read_bvh(prt_context* c, prt_bvh_node* nodes, uint32_t* order)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_read_bvh: no acceleration structure");
    if(c->twoLevel()) return fail(c, PRT_E_INVALID, "prt_read_bvh: two-level scene, use prt_read_blas_bvh");
    return read_tree(c, c->top, nodes, order);
}

int prt_read_blas_bvh(prt_context* c, uint32_t k, prt_bvh_node* nodes, uint32_t* order, uint32_t* nNodes)
{
    if(!c) return PRT_E_INVALID;
    if(!c->haveAccel) return fail(c, PRT_E_NOSCENE, "prt_read_blas_bvh: no acceleration structure");
    if(k >= c->blas.size()) return fail(c, PRT_E_INVALID, "prt_read_blas_bvh: structure %u of %zu", k, c->blas.size());
    if(nNodes) *nNodes = c->blas[k].nNodes;
    return read_tree(c, c->blas[k], nodes, order);
}

// measurement builds (-DPRT_SHADE_STAMPS): the raw statistics words
#ifdef PRT_TIMELINE
// measurement builds only (tools/timeline.py): the per-wave time stamps of the last traversal stream that ran
int prt_debug_read_timeline(prt_context* c, unsigned long long* out, unsigned nwords)
{
    if(!c || !out || nwords > 8192u * 4u) return PRT_E_INVALID;
    if(hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return PRT_E_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)nwords * 8) == hipSuccess ? PRT_OK : PRT_E_HIP;
}
#endif
int prt_debug_read_stats(prt_context* c, unsigned long long* out48)
{
    if(!c || !out48) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    HIPCHK(c, hipMemcpy(out48, c->stats, kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_read_sobol(prt_context* c, uint32_t* out)
{
    if(!c || !out) return PRT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->dSobol, 64 * 32 * 4, hipMemcpyDeviceToHost));
    return PRT_OK;
}

}  // extern "C"
