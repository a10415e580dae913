// prt_group.hip — several GPUs behind one handle, inside one process: the sample-sharded multi-GPU form of the
// path for C / C++ callers (SURVEY.md 8b "Multi-GPU (new)", 8e). No counterpart in the reference (one VkDevice, one
// queue). A group is a layer over the public prt_* entry points: one prt_context per member, one host thread per
// member while rendering, iterations dealt out round-robin (member r renders first+r, first+r+G, ...), and ONE
// exchange step: the sum of the members' float4 images into the root's image.
//   * distinct devices: RCCL ncclReduce(sum, float, W*H*4) over xGMI, one communicator per member
//     (ncclCommInitAll), loaded at run time from librccl.so so that single-GPU users do not need it;
//   * a device listed twice (tests on a one-GPU box, oversubscription) or PRT_GROUP_NO_RCCL=1: peer copies into a
//     staging buffer on the root's device plus an add kernel, in member order.
// Python / multi-process callers use torch.distributed instead (pathtracer_amd/sharding.py, bench.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/prt.h"

namespace {

__global__ void __launch_bounds__(256) k_add_image(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const float4 a = dst[i], b = src[i];
    dst[i]         = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// the few RCCL entry points used, resolved with dlsym (types as in rccl.h: opaque comm, int enums)
struct Rccl
{
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist)                                        = nullptr;
    int (*CommDestroy)(void* comm)                                                                        = nullptr;
    int (*GroupStart)()                                                                                   = nullptr;
    int (*GroupEnd)()                                                                                     = nullptr;
    int (*Reduce)(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, hipStream_t s) = nullptr;
    const char* (*GetErrorString)(int)                                                                    = nullptr;
    bool load()
    {
        if(lib) return true;
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if(!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if(!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if(!lib) return false;
        CommInitAll    = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy    = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart     = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd       = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Reduce         = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Reduce && GetErrorString;
    }
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

thread_local std::string g_createError;

}  // namespace

struct prt_group
{
    std::vector<prt_context*> ctx;
    std::vector<int>          device;
    std::vector<hipStream_t>  stream;  // one per member, on the member's device (reduce traffic)
    std::vector<void*>        comm;    // RCCL communicators, empty when the copy path is used
    Rccl                      rccl;
    uint32_t                  w = 0, h = 0;
    float4*                   staging = nullptr;  // copy path: W*H float4 on the root's device
    int                       stagingDevice = -1;
    std::string               err;
    bool                      useRccl = false;
};

static int gfail(prt_group* g, int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if(g)
        g->err = buf;
    else
        g_createError = buf;
    return code;
}

// run f(member index) on one host thread per member; first non-zero return code wins
template <class F>
static int for_members(prt_group* g, F f)
{
    const size_t     n = g->ctx.size();
    std::vector<int> rc(n, 0);
    if(n == 1)
        rc[0] = f(0);
    else
    {
        std::vector<std::thread> th;
        for(size_t i = 0; i < n; ++i) th.emplace_back([&, i] { rc[i] = f((int)i); });
        for(std::thread& t : th) t.join();
    }
    for(size_t i = 0; i < n; ++i)
        if(rc[i]) return gfail(g, rc[i], "member %zu (device %d): %s", i, g->device[i], prt_last_error(g->ctx[i]));
    return PRT_OK;
}

extern "C" {

int prt_group_create(int n, const int* ordinals, uint32_t width, uint32_t height, prt_group** out)
{
    if(!out || !ordinals || n < 1 || n > 64) return gfail(nullptr, PRT_E_INVALID, "prt_group_create: 1..64 devices");
    *out         = nullptr;
    prt_group* g = new prt_group;
    g->w         = width;
    g->h         = height;
    bool distinct = true;
    for(int i = 0; i < n; ++i)
        for(int j = 0; j < i; ++j)
            if(ordinals[i] == ordinals[j]) distinct = false;
    for(int i = 0; i < n; ++i)
    {
        prt_context* c  = nullptr;
        const int    rc = prt_create(ordinals[i], width, height, &c);
        if(rc != PRT_OK)
        {
            gfail(nullptr, rc, "prt_group_create: device %d: %s", ordinals[i], prt_last_error(nullptr));
            prt_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(c);
        g->device.push_back(ordinals[i]);
        hipStream_t s = nullptr;
        if(hipSetDevice(ordinals[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        {
            gfail(nullptr, PRT_E_HIP, "prt_group_create: cannot create a stream on device %d", ordinals[i]);
            prt_group_destroy(g);
            return PRT_E_HIP;
        }
        g->stream.push_back(s);
    }
    const char* no = getenv("PRT_GROUP_NO_RCCL");
    const char* force = getenv("PRT_GROUP_FORCE_RCCL");  // also for a single member (exercises the RCCL plumbing)
    const bool  forced = force && *force && *force != '0';
    if(distinct && (n > 1 || forced) && !(no && *no && *no != '0'))
    {
        if(!g->rccl.load())
        {
            gfail(nullptr, PRT_E_HIP, "prt_group_create: librccl.so cannot be loaded (%s)", dlerror());
            prt_group_destroy(g);
            return PRT_E_HIP;
        }
        g->comm.assign((size_t)n, nullptr);
        const int rc = g->rccl.CommInitAll(g->comm.data(), n, ordinals);
        if(rc != 0)
        {
            gfail(nullptr, PRT_E_HIP, "prt_group_create: ncclCommInitAll: %s", g->rccl.GetErrorString(rc));
            g->comm.clear();
            prt_group_destroy(g);
            return PRT_E_HIP;
        }
        g->useRccl = true;
    }
    *out = g;
    return PRT_OK;
}

int prt_group_destroy(prt_group* g)
{
    if(!g) return PRT_E_INVALID;
    for(void* c : g->comm)
        if(c) g->rccl.CommDestroy(c);
    if(g->staging)
    {
        (void)hipSetDevice(g->stagingDevice);
        (void)hipFree(g->staging);
    }
    for(size_t i = 0; i < g->stream.size(); ++i)
    {
        (void)hipSetDevice(g->device[i]);
        (void)hipStreamDestroy(g->stream[i]);
    }
    for(prt_context* c : g->ctx) prt_destroy(c);
    delete g;
    return PRT_OK;
}

const char* prt_group_last_error(const prt_group* g) { return g ? g->err.c_str() : g_createError.c_str(); }
int         prt_group_size(const prt_group* g) { return g ? (int)g->ctx.size() : 0; }
int         prt_group_uses_rccl(const prt_group* g) { return g && g->useRccl ? 1 : 0; }
prt_context* prt_group_context(prt_group* g, int member)
{
    return (g && member >= 0 && (size_t)member < g->ctx.size()) ? g->ctx[(size_t)member] : nullptr;
}

int prt_group_set_scene(prt_group* g, const prt_vertex* v, size_t nv, const uint32_t* idx, size_t nidx,
                        const prt_material* m, size_t nm, const prt_texture* tex, size_t ntex)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) {
        const int rc = prt_set_scene(g->ctx[(size_t)i], v, nv, idx, nidx, m, nm, tex, ntex);
        return rc ? rc : prt_build_accel(g->ctx[(size_t)i]);
    });
}
int prt_group_set_instances(prt_group* g, const prt_blas_range* blas, size_t nblas, const prt_instance* inst, size_t ninst)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) {
        const int rc = prt_set_instances(g->ctx[(size_t)i], blas, nblas, inst, ninst);
        return rc ? rc : prt_build_accel(g->ctx[(size_t)i]);
    });
}
int prt_group_set_scramble_seed(prt_group* g, uint32_t seed)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) { return prt_set_scramble_seed(g->ctx[(size_t)i], seed); });
}
int prt_group_set_uniforms(prt_group* g, const prt_uniforms* ubo)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) { return prt_set_uniforms(g->ctx[(size_t)i], ubo); });
}
int prt_group_set_flags(prt_group* g, uint32_t flags)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) { return prt_set_flags(g->ctx[(size_t)i], flags); });
}
int prt_group_clear_accum(prt_group* g)
{
    if(!g) return PRT_E_INVALID;
    return for_members(g, [&](int i) { return prt_clear_accum(g->ctx[(size_t)i]); });
}

int prt_group_render(prt_group* g, int mode, uint32_t first, uint32_t n_iter)
{
    if(!g) return PRT_E_INVALID;
    const uint32_t G = (uint32_t)g->ctx.size();
    return for_members(g, [&](int i) {
        const uint32_t r     = (uint32_t)i;
        const uint32_t count = n_iter > r ? (n_iter - r + G - 1) / G : 0;
        if(count == 0) return (int)PRT_OK;
        const int rc = prt_render(g->ctx[r], mode, first + r, count, G);
        return rc ? rc : prt_sync(g->ctx[r]);
    });
}

int prt_group_reduce(prt_group* g, int root)
{
    if(!g || root < 0 || (size_t)root >= g->ctx.size()) return g ? gfail(g, PRT_E_INVALID, "prt_group_reduce: bad root") : PRT_E_INVALID;
    const size_t G = g->ctx.size();
    const size_t npix = (size_t)g->w * g->h;
    std::vector<void*> accum(G, nullptr);
    for(size_t i = 0; i < G; ++i)
    {
        int rc = prt_sync(g->ctx[i]);
        if(!rc) rc = prt_device_accum_ptr(g->ctx[i], &accum[i]);
        if(rc) return gfail(g, rc, "prt_group_reduce: member %zu: %s", i, prt_last_error(g->ctx[i]));
    }
    if(G == 1 && !g->useRccl) return PRT_OK;
#define GHIP(expr)                                                                                          \
    do                                                                                                      \
    {                                                                                                       \
        hipError_t e_ = (expr);                                                                             \
        if(e_ != hipSuccess) return gfail(g, PRT_E_HIP, "prt_group_reduce: %s: %s", #expr, hipGetErrorString(e_)); \
    } while(0)
    if(g->useRccl)
    {
        int rc = g->rccl.GroupStart();
        for(size_t i = 0; i < G && rc == 0; ++i)
        {
            GHIP(hipSetDevice(g->device[i]));
            rc = g->rccl.Reduce(accum[i], accum[i], npix * 4, kNcclFloat32, kNcclSum, root, g->comm[i], g->stream[i]);
        }
        const int rc2 = g->rccl.GroupEnd();
        if(rc || rc2) return gfail(g, PRT_E_HIP, "prt_group_reduce: ncclReduce: %s", g->rccl.GetErrorString(rc ? rc : rc2));
        for(size_t i = 0; i < G; ++i)
        {
            GHIP(hipSetDevice(g->device[i]));
            GHIP(hipStreamSynchronize(g->stream[i]));
        }
    }
    else
    {
        const int rd = g->device[(size_t)root];
        GHIP(hipSetDevice(rd));
        if(g->staging && g->stagingDevice != rd)
        {
            GHIP(hipSetDevice(g->stagingDevice));
            GHIP(hipFree(g->staging));
            g->staging = nullptr;
            GHIP(hipSetDevice(rd));
        }
        if(!g->staging)
        {
            GHIP(hipMalloc((void**)&g->staging, npix * sizeof(float4)));
            g->stagingDevice = rd;
        }
        hipStream_t s = g->stream[(size_t)root];
        for(size_t i = 0; i < G; ++i)
        {
            if((int)i == root) continue;
            GHIP(hipMemcpyPeerAsync(g->staging, rd, accum[i], g->device[i], npix * sizeof(float4), s));
            hipLaunchKernelGGL(k_add_image, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (float4*)accum[(size_t)root],
                               (const float4*)g->staging, (uint32_t)npix);
            GHIP(hipGetLastError());
        }
        GHIP(hipStreamSynchronize(s));
    }
#undef GHIP
    // the root now holds the total; the others start their next partial sums from zero
    for(size_t i = 0; i < G; ++i)
        if((int)i != root)
        {
            const int rc = prt_clear_accum(g->ctx[i]);
            if(rc) return gfail(g, rc, "prt_group_reduce: member %zu: %s", i, prt_last_error(g->ctx[i]));
        }
    return PRT_OK;
}

int prt_group_read_accum(prt_group* g, int root, float* rgba_out)
{
    if(!g || root < 0 || (size_t)root >= g->ctx.size()) return PRT_E_INVALID;
    const int rc = prt_read_accum(g->ctx[(size_t)root], rgba_out);
    return rc ? gfail(g, rc, "prt_group_read_accum: %s", prt_last_error(g->ctx[(size_t)root])) : PRT_OK;
}

int prt_group_get_stats(prt_group* g, prt_stats* out)
{
    if(!g || !out) return PRT_E_INVALID;
    memset(out, 0, sizeof *out);
    for(size_t i = 0; i < g->ctx.size(); ++i)
    {
        prt_stats s;
        const int rc = prt_get_stats(g->ctx[i], &s);
        if(rc) return gfail(g, rc, "prt_group_get_stats: member %zu: %s", i, prt_last_error(g->ctx[i]));
        out->rays += s.rays;
        out->iterations += s.iterations;
        out->trace_overruns = s.trace_overruns > 0xffffffffu - out->trace_overruns ? 0xffffffffu : out->trace_overruns + s.trace_overruns;
        if(i == 0)
        {
            out->bvh_nodes       = s.bvh_nodes;
            out->bvh_inner_nodes = s.bvh_inner_nodes;
            out->bvh_depth       = s.bvh_depth;
            out->bvh_build_ms    = s.bvh_build_ms;
            out->bvh_refit_ms    = s.bvh_refit_ms;
        }
    }
    return PRT_OK;
}

}  // extern "C"
