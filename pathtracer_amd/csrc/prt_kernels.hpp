// prt_kernels.hpp — the pathRT / AO integrators as wavefront kernels for gfx950.
//
// The reference runs one megakernel ray-gen invocation per pixel (shaders/pathRT.rgen:621-974,
// shaders/AO.rgen:183-273) and leaves divergence to the RT hardware's scheduler. Here one
// iteration is a fixed sequence of launches with no host round trip:
//
//   generate -> [ extend -> shade -> shadow+connect ] x (numIndirectBounces + 1)
//
//   generate  : Sobol pixel jitter, primary ray                         (pathRT.rgen:627-690)
//   extend    : closest-hit traversal of the queued rays                (traceNV :692, :963)
//   shade     : vertex/material fetch, textures, bump, light sample, GGX VNDF sample, termination;
//               emits the shadow ray + both possible radiance terms and the next ray;
//               surviving paths are compacted into the next queue with wave ballots  (:699-961)
//   shadow    : any-hit traversal; the vertex's radiance term becomes Ei_vis when the light sample is
//               visible (it was stored as Ei_occ by shade)                            (:818-847)
//   resolve   : E += term[0] + term[1] + ... in path order, once per pass             (:939-942)
//
// Passes (iterations) are independent until `resolve`, so several of them are in flight on separate
// HIP streams (prt_api.hip); `resolve` runs in iteration order, which keeps the fp32 sums bit-exact.
//
// Per-path state lives at QUEUE SLOTS (FrameView::gen: SoA float4 records, two generations; k_shade writes the
// survivors of a vertex compacted into the next generation), only the radiance terms are stored by pixel. Queue
// lengths live in device memory, kernels are persistent (wave-granular work fetch), so the launch sequence is
// identical every iteration.
#pragma once
#include "prt_trace.hpp"

namespace prt {

#define PRT_FLAG_LITERAL 1u
#define PRT_FLAG_RUSSIAN_ROULETTE 2u

struct SceneView
{
    const float4*       stri;  // 7 x float4 per triangle: p,n,t of the three corners + material id of corner 1
    const uint32_t*     sobolBytes;  // [64][4][256]
    const prt_vertex*   verts;
    const uint32_t*     idx;
    const prt_material* mats;
    const TexDesc*      tex;
    const uint32_t*     sobol;     // [64][32]
    const uint32_t*     scrTable;  // optional [32][npix]
    uint32_t            nmats, ntex, seed, ntri;
    BvhView             bvh;
    // two-level scenes (kernels instantiated with TWO): instances in the caller's order
    const float4*       shadeInst;  // 8 x float4 per instance (ShadeInst below)
    const uint32_t*     instBase;   // flattened primitive id of every instance's first triangle (ninst + 1 entries)
    const uint32_t*     instLookup; // instance that holds primitive id k << instShift (prt_api.hip: instance_tables)
    uint32_t            instShift;
    uint32_t            ninst;
};

struct PathGen
{
    float4*   rayO;  // Ro.xyz, tmin
    float4*   rayD;  // Rd.xyz, tmax
    float4*   hit;   // prim bits, u, v, t
    float4*   thr;   // throughput.xyz, p
    uint32_t* px;    // pixel of the pass
};

struct FrameView
{
    uint32_t  w, h;
    uint32_t  npix;        // pixels of this pass: npixImg times the number of iterations batched into it; "pixel" k
                           // of the pass is image pixel k % npixImg of iteration ubo.iteration + (k / npixImg) * iterStride
    uint32_t  npixImg, iterStride;
    float4*   accum;
    // Path state lives at QUEUE SLOTS, not at pixels: the paths that survive vertex b are written, compacted, into
    // generation (b+1)&1 by k_shade(b), traced there by k_shadow_extend(b) and read back by k_shade(b+1) — every
    // access of a wave is a run of consecutive records. Generation 0 at vertex 0 is the identity (slot == pixel of
    // the pass, `px` unused).
    PathGen   gen[2];
    // shadow rays of the current vertex, by slot of the shadow queue
    float4*   shO;    // hitPoint.xyz, tmin
    float4*   shD;    // vLight.xyz, -
    float4*   eiVis;  // radiance term if the light sample is visible (w = color.w)
    uint32_t* shPx;   // pixel of the pass the term belongs to
    float4*   contrib;  // [vertex][npix]: radiance term of every path vertex of this pass (w = color.w)
    float2*   aaOff;  // rayOffset (only read when numAArays != 1)
    uint32_t* hitIds; // primary-ray primitive id per pixel
    uint8_t*  dead;   // pixel returned early in this iteration (primary miss)
    uint8_t*  nvert;  // terms stored for the pixel in this pass | 0x40 overwrite E | 0x80 reset E first
    unsigned long long* stats;  // [0] closest-hit rays, [1] any-hit rays, [2..3] nodes / triangles visited by
                                // closest-hit rays, [4..5] by any-hit rays, [8..13] / [14..19] wave-step utilisation, [20..23] / [24..27] phase ticks (profiling level 2 only), [28] waves that hit the safety bound of trace_stream
};

PRT_DEV uint32_t wave_push(bool pred, uint32_t* counter)
{
    const unsigned long long mask = __ballot(pred);
    const uint32_t           n    = (uint32_t)__popcll(mask);
    const int                lane = threadIdx.x & 63;
    uint32_t                 base = 0;
    if(n)
    {
        const int leader = __ffsll((long long)mask) - 1;
        if(lane == leader) base = atomicAdd(counter, n);
        base = (uint32_t)__shfl((int)base, leader, 64);
    }
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Block-wide variant (all 256 threads call it): one atomic per block instead of one per wave — the
// counters are single addresses, and same-address atomics serialise at the memory side.
PRT_DEV uint32_t block_push(bool pred, uint32_t* counter, uint32_t* lds /*[8]*/)
{
    const unsigned long long mask = __ballot(pred);
    const int                lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if(lane == 0) lds[wid] = (uint32_t)__popcll(mask);
    __syncthreads();
    if(threadIdx.x == 0)
    {
        const uint32_t n0 = lds[0], n1 = lds[1], n2 = lds[2], n3 = lds[3];
        const uint32_t tot = n0 + n1 + n2 + n3;
        const uint32_t b   = tot ? atomicAdd(counter, tot) : 0u;
        lds[4] = b;
        lds[5] = b + n0;
        lds[6] = b + n0 + n1;
        lds[7] = b + n0 + n1 + n2;
    }
    __syncthreads();
    const uint32_t base = lds[4 + wid];
    __syncthreads();
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// block_push in two halves, so that the round trip of the counter's atomic (it executes at the memory side) runs
// beside the work between them. begin: every wave leaves its count in lds[0..3], thread 0 issues the atomic and KEEPS
// the returned base in a register (no wait here: the wait lands where the value is first used, in push_end).
// end: thread 0 turns the base into per-wave bases (lds[4..7]); every thread gets its slot. `lds` = 8 words that no
// other push uses between the two calls.
struct PushTicket
{
    unsigned long long mask;
    uint32_t           base;  // thread 0 only
};
#ifndef PRT_SHADE_BLOCK
// Threads per block of k_shade = path vertices per pair of queue atomics (the compaction is block-wide: one atomicAdd on
// the shadow queue's length and one on the next generation's per trip). Those two words are what bounded the kernel at
// 256 threads: 1.04 M same-address atomics per launch of 132.7 M slots = 111 per microsecond and word, beyond the ~88
// a word sustains (MI355X_MICROARCH.md, dequeue). Measured per launch (bench, 64 iterations per pass): 128 threads
// 8.66 ms, 256: 4.67, 512: 4.39, 1024: 4.63 (one block per CU: the barriers span the CU).
#define PRT_SHADE_BLOCK 512
#endif
#define PRT_SHADE_NW (PRT_SHADE_BLOCK / 64)
PRT_DEV PushTicket block_push_begin(bool pred, uint32_t* counter, uint32_t* lds /*[2 * PRT_SHADE_NW]*/)
{
    PushTicket t;
    t.mask = __ballot(pred);
    t.base = 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if(lane == 0) lds[wid] = (uint32_t)__popcll(t.mask);
    __syncthreads();
    if(threadIdx.x == 0)
    {
        uint32_t tot = 0;
        for(int w = 0; w < PRT_SHADE_NW; ++w) tot += lds[w];
#ifdef PRT_FAKE_PUSH  // timing builds only (wrong queues): what k_shade costs without the same-address atomics
        t.base = blockIdx.x * PRT_SHADE_BLOCK;
#else
        if(tot) t.base = atomicAdd(counter, tot);
#endif
    }
    return t;
}
// thread 0's part of the second half (call for every pending ticket, then ONE __syncthreads, then block_push_slot)
PRT_DEV void block_push_publish(const PushTicket& t, uint32_t* lds)
{
    if(threadIdx.x == 0)
    {
        uint32_t run = t.base;
        for(int w = 0; w < PRT_SHADE_NW; ++w)
        {
            lds[PRT_SHADE_NW + w] = run;
            run += lds[w];
        }
    }
}
PRT_DEV uint32_t block_push_slot(const PushTicket& t, const uint32_t* lds)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    return lds[PRT_SHADE_NW + wid] + (uint32_t)__popcll(t.mask & ((1ull << lane) - 1ull));
}

PRT_DEV uint32_t wave_fetch(uint32_t* workCtr, uint32_t n)
{
    uint32_t base = 0;
    if((threadIdx.x & 63) == 0) base = atomicAdd(workCtr, n);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
}

// Per-lane tallies are kept in registers for the whole persistent loop and folded once per wave at the
// end (all lanes have left the loop together, so the wave is converged for the shuffles).
PRT_DEV uint32_t wave_sum(uint32_t v)
{
    for(int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
// `queued` = length of the launch's queue, `rays` = rays this lane loaded that count in the statistics. The total
// is added once per launch; a wave only touches the counter for the (rare) placeholder rays it must take off
// again — one same-address atomic per wave at the end of every launch is a measurable tail otherwise.
template <bool INSTR>
PRT_DEV void flush_counts(const FrameView& f, uint32_t queued, uint32_t skippedLane, TraceCount tc, int kind)
{
    const uint32_t skipped = wave_sum(skippedLane);
    uint32_t       n = 0, t = 0;
    if(INSTR)
    {
        n = wave_sum(tc.nodes);
        t = wave_sum(tc.tris);
    }
    if(blockIdx.x == 0 && threadIdx.x == 0 && queued) atomicAdd(f.stats + kind, (unsigned long long)queued);
    if(tc.overrun && (threadIdx.x & 63) == 0) atomicAdd(f.stats + 28, 1ull);  // never expected: see trace_stream
    if((threadIdx.x & 63) == 0)
    {
        if(skipped) atomicAdd(f.stats + kind, ~(unsigned long long)skipped + 1ull);
        if(INSTR)
        {
            atomicAdd(f.stats + 2 + 2 * kind, (unsigned long long)n);
            atomicAdd(f.stats + 3 + 2 * kind, (unsigned long long)t);
            unsigned long long* u = f.stats + 8 + 6 * kind;
            atomicAdd(u + 0, (unsigned long long)tc.wNode);
            atomicAdd(u + 1, (unsigned long long)tc.lNode);
            atomicAdd(u + 2, (unsigned long long)tc.wTri);
            atomicAdd(u + 3, (unsigned long long)tc.lTri);
            atomicAdd(u + 4, (unsigned long long)tc.wSetup);
            atomicAdd(u + 5, (unsigned long long)tc.lSetup);
            unsigned long long* y = f.stats + 20 + 4 * kind;
            atomicAdd(y + 0, (unsigned long long)tc.cSetup);
            atomicAdd(y + 1, (unsigned long long)tc.cNode);
            atomicAdd(y + 2, (unsigned long long)tc.cTri);
            atomicAdd(y + 3, (unsigned long long)tc.cTotal);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// generate — pathRT.rgen:624-690
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_generate(SceneView sc, FrameView f, const prt_uniforms ubo_, uint32_t flags,
                                                  uint32_t aa)
{
    const prt_uniforms* ubo = &ubo_;
    const uint32_t      px  = blockIdx.x * blockDim.x + threadIdx.x;
    if(px >= f.npix) return;
    const bool literal = (flags & PRT_FLAG_LITERAL) != 0;
    f.nvert[px] = (aa == 0 && literal && !(ubo->iteration > 1u)) ? 0x80 : 0;  // pathRT.rgen:649-654
    if(aa == 0)
    {
        f.dead[px] = 0;
    }
    else if(f.dead[px])
    {
        f.gen[0].rayO[px] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        f.gen[0].rayD[px] = make_float4(0.0f, 0.0f, 1.0f, -1.0f);  // empty interval: traces as a miss
        return;
    }
    const uint32_t sub = px / f.npixImg, ip = px - sub * f.npixImg;  // iteration of the batch, image pixel
    const uint32_t x = ip % f.w, y = ip / f.w;
    const uint32_t s0 = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 0);
    const uint32_t s1 = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 1);
    const uint32_t sobolIndex = s0 / 2u + (ubo->iteration + sub * f.iterStride) + aa;
    f2 off;
    off.x = sobol1D(sc.sobol, sobolIndex, 2u, s0);
    off.y = sobol1D(sc.sobol, sobolIndex, 3u, s1);
    f3 Ro, Rd;
    if(ubo->numAArays == 1)
    {
        off.x = off.x * 2.0f - 1.0f;
        off.y = off.y * 2.0f - 1.0f;
        off.x = off.x * ubo->filterRadius;
        off.y = off.y * ubo->filterRadius;
        primary_ray(ubo->projViewInverse, x, y, f.w, f.h, off.x + 0.5f, off.y + 0.5f, Ro, Rd);
    }
    else
    {
        primary_ray(ubo->projViewInverse, x, y, f.w, f.h, off.x, off.y, Ro, Rd);
        f.aaOff[px] = make_float2(off.x, off.y);
    }
    st_nt(f.gen[0].rayO + px, make_float4(Ro.x, Ro.y, Ro.z, 0.000001f));
    st_nt(f.gen[0].rayD + px, make_float4(Rd.x, Rd.y, Rd.z, 1.0f));
    // throughput (1,1,1) and p = 1 of a primary ray are not stored: k_shade(0) knows them
}

// AO.rgen:185-205 — pixel-centre primary ray, tmin 1e-5
__global__ void __launch_bounds__(256) k_ao_generate(FrameView f, const prt_uniforms ubo_, uint32_t flags)
{
    const prt_uniforms* ubo = &ubo_;
    const uint32_t      px  = blockIdx.x * blockDim.x + threadIdx.x;
    if(px >= f.npix) return;
    if((flags & PRT_FLAG_LITERAL) && !(ubo->iteration > 1u)) f.accum[px] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // one iteration per pass
    f3 Ro, Rd;
    const uint32_t ip = px % f.npixImg;  // several iterations per pass: pixel px of the pass = image pixel ip of iteration px / npixImg
    primary_ray(ubo->projViewInverse, ip % f.w, ip / f.w, f.w, f.h, 0.5f, 0.5f, Ro, Rd);
    f.gen[0].rayO[px] = make_float4(Ro.x, Ro.y, Ro.z, 0.00001f);
    f.gen[0].rayD[px] = make_float4(Rd.x, Rd.y, Rd.z, 1.0f);
}

// ---------------------------------------------------------------------------------------------
// extend — closest hit for every ray of one generation of path state (slots 0 .. count-1)
// ---------------------------------------------------------------------------------------------
struct ExtendIO
{
    struct
    {
        const float4 *rayO, *rayD;
        float4*       hit;
    } g;
    uint32_t* hitIds;  // primary rays only (slot == pixel), else nullptr
    PRT_DEV ExtendIO(const PathGen& p, uint32_t* ids) : g{p.rayO, p.rayD, p.hit}, hitIds(ids) {}
    PRT_DEV bool load(uint32_t i, f3& o, float& tmin, f3& d, float& tmax, uint32_t& tag) const
    {
        const float4 oq = ld_nt(g.rayO + i), dq = ld_nt(g.rayD + i);
        o    = mk3(oq.x, oq.y, oq.z);
        tmin = oq.w;
        d    = mk3(dq.x, dq.y, dq.z);
        tmax = dq.w;
        tag  = i;
        return !(dq.w < 0.0f);  // placeholder rays of pixels that already returned are not counted
    }
    PRT_DEV void store(uint32_t i, const Hit& h) const
    {
        st_res(g.hit + i, make_float4(__int_as_float((int)h.prim), h.u, h.v, h.t));
        if(hitIds) st_nt(hitIds + i, h.prim);
    }
};

#ifndef PRT_TRACE_ATTR
#define PRT_TRACE_ATTR
#endif
#ifndef PRT_TRACE_WAVES_TWO
#define PRT_TRACE_WAVES_TWO 6  // waves per SIMD of the two-level traversal kernels (80 VGPRs since the register diet of round 3: 5: 3300, 6: 3510 Mrays/s on the 2 001-instance scene)
#endif
template <bool INSTR, bool TWO = false>
__global__ void __launch_bounds__(PRT_TRACE_BLOCK, INSTR ? 4 : (TWO ? PRT_TRACE_WAVES_TWO : PRT_TRACE_WAVES)) PRT_TRACE_ATTR k_extend(SceneView sc, FrameView f, uint32_t generation,
                                                            const uint32_t* __restrict__ countPtr,
                                                            uint32_t* workCtr, uint32_t primary, TraceTune tune)
{
    __shared__ __align__(16) uint32_t stackLds[TWO ? PRT_TRACE_LDS_WORDS_TWO : PRT_TRACE_LDS_WORDS];
    const uint4* topLds = stage_top_nodes<PRT_LS(TWO)>(sc.bvh, stackLds);
    const uint32_t      count = countPtr ? *countPtr : f.npix;
    uint32_t            nskip = 0;
    TraceCount          tc    = {};
    ExtendIO            io(f.gen[generation & 1u], primary ? f.hitIds : nullptr);
    trace_stream<false, INSTR, ExtendIO, TWO>(sc.bvh, &stackLds[threadIdx.x], topLds, count, workCtr, tune, io, nskip, tc);
    flush_counts<INSTR>(f, count, nskip, tc, 0);
}

// ---------------------------------------------------------------------------------------------
// shade — pathRT.rgen:699-961 for one path vertex
// ---------------------------------------------------------------------------------------------
struct VertexU
{
    f3 pos, normal;
    f2 uv;
};
// Shading record of one triangle, gathered once at prt_set_scene from the vertex/index buffers
// (unpackVertex, pathRT.rgen:94-119): corner k = rows 2k, 2k+1 = (p.xyz, n.x), (n.y, n.z, t.x, t.y);
// row 6.x = materialID of corner 1 (the shader takes the material from v1, :712).
#ifndef PRT_SHADE_TRI_F4
#define PRT_SHADE_TRI_F4 8  // pitch in float4: one aligned 128-byte line per triangle (7 rows used)
#endif
struct TriU
{
    VertexU v0, v1, v2;
    int32_t mat;
};
PRT_DEV VertexU unpack_corner(float4 a, float4 b)
{
    VertexU v;
    v.pos    = mk3(a.x, a.y, a.z);
    v.normal = mk3(a.w, b.x, b.y);
    v.uv.x   = b.z;
    v.uv.y   = b.w;
    return v;
}
PRT_DEV TriU load_tri(const float4* __restrict__ stri, uint32_t prim)
{
    const float4* q  = stri + (size_t)prim * PRT_SHADE_TRI_F4;
    const float4  q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
    TriU          t;
    t.v0  = unpack_corner(q0, q1);
    t.v1  = unpack_corner(q2, q3);
    t.v2  = unpack_corner(q4, q5);
    t.mat = __float_as_int(q6.x);
    return t;
}
// Two-level scenes: the flattened primitive `prim` belongs to the last instance whose base is <= prim (found through
// the bucket table: the search runs over the instances that share the id's bucket); its triangle
// record (object space) is taken to world space the way an instanced scene is flattened (host/Scene.cpp, oracle.cpp
// shadingVertex): p = (M0*x + M1*y) + (M2*z + M3), n = normalize(N*n) with N = transpose(inverse(M)), a zero normal
// stays zero, an identity instance is taken bit for bit.
// ShadeInst record, 128 bytes: rows 0-2 object -> world (row i = (M[0][i], M[1][i], M[2][i], M[3][i])), rows 3-5 the
// same rows of N (.w unused), row 6 = (first triangle of the bottom-level structure, flags: 1 = identity, -, -).
PRT_DEV VertexU xform_corner(VertexU v, float4 m0, float4 m1, float4 m2, float4 n0, float4 n1, float4 n2)
{
    const f3 p = v.pos, n = v.normal;
    v.pos = mk3((m0.x * p.x + m0.y * p.y) + (m0.z * p.z + m0.w), (m1.x * p.x + m1.y * p.y) + (m1.z * p.z + m1.w),
                (m2.x * p.x + m2.y * p.y) + (m2.z * p.z + m2.w));
    f3 nn = mk3((n0.x * n.x + n0.y * n.y) + n0.z * n.z, (n1.x * n.x + n1.y * n.y) + n1.z * n.z,
                (n2.x * n.x + n2.y * n.y) + n2.z * n.z);
    if(dot3(nn, nn) > 0.0f) nn = normalize3(nn);
    v.normal = nn;
    return v;
}
PRT_DEV TriU load_tri_inst(const SceneView& sc, uint32_t prim)
{
    const uint32_t bucket = prim >> sc.instShift;
    uint32_t       lo = sc.instLookup[bucket], hi = sc.instLookup[bucket + 1u] + 1u;
    while(hi - lo > 1u)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if(sc.instBase[mid] <= prim)
            lo = mid;
        else
            hi = mid;
    }
    const float4*  rec   = sc.shadeInst + (size_t)lo * 8;
    const float4   mt    = rec[6];
    const uint32_t local = prim - sc.instBase[lo] + (uint32_t)__float_as_int(mt.x);
    TriU           t     = load_tri(sc.stri, local);
    if(!((uint32_t)__float_as_int(mt.y) & 1u))
    {
        const float4 m0 = rec[0], m1 = rec[1], m2 = rec[2], n0 = rec[3], n1 = rec[4], n2 = rec[5];
        t.v0 = xform_corner(t.v0, m0, m1, m2, n0, n1, n2);
        t.v1 = xform_corner(t.v1, m0, m1, m2, n0, n1, n2);
        t.v2 = xform_corner(t.v2, m0, m1, m2, n0, n1, n2);
    }
    return t;
}
__global__ void __launch_bounds__(256) k_gather_shading(const prt_vertex* __restrict__ verts,
                                                        const uint32_t* __restrict__ idx, uint32_t ntri,
                                                        float4* __restrict__ stri)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= ntri) return;
    float4* o = stri + (size_t)i * PRT_SHADE_TRI_F4;
    for(int k = 0; k < 3; ++k)
    {
        const prt_vertex& v = verts[idx[3 * i + k]];
        o[2 * k]            = make_float4(v.p[0], v.p[1], v.p[2], v.n[0]);
        o[2 * k + 1]        = make_float4(v.n[1], v.n[2], v.t[0], v.t[1]);
    }
    o[6] = make_float4(__int_as_float(verts[idx[3 * i + 1]].materialID), 0.0f, 0.0f, 0.0f);
}

#ifndef PRT_SHADE_WAVES_TWO
#define PRT_SHADE_WAVES_TWO 4  // two-level scenes (2: 20.3 ms, 4: 19.3 ms per pass)
#endif
#ifndef PRT_SHADE_WAVES
#define PRT_SHADE_WAVES 4  // 117 VGPRs, no scratch (tools/kernel_resources.py); 5 or 6 spill and are no faster, 2-3 leave the compiler 132 VGPRs = 3 waves per SIMD
#endif
template <bool TWO = false>
__global__ void __launch_bounds__(PRT_SHADE_BLOCK, TWO ? PRT_SHADE_WAVES_TWO : PRT_SHADE_WAVES) k_shade(SceneView sc, FrameView f, const prt_uniforms ubo_, uint32_t flags,
                                               uint32_t bounce, const uint32_t* __restrict__ countIn,
                                               uint32_t* countNext, uint32_t* countShadow)
{
    const PathGen gin = f.gen[bounce & 1u], gout = f.gen[(bounce + 1u) & 1u];
    const prt_uniforms* ubo = &ubo_;
    // Sobol byte tables of this vertex's dimensions (4 + 4*bounce .. +3) and of the roulette dimension
    __shared__ uint32_t sobT[5 * 1024];
    {
        const uint32_t dim0 = 4u + 4u * bounce, dimR = 48u + bounce + 1u;
        for(uint32_t k = threadIdx.x; k < 5u * 1024u; k += PRT_SHADE_BLOCK)
        {
            const uint32_t d = k >> 10;
            const uint32_t g = d < 4u ? dim0 + d : dimR;
            sobT[k]          = g < 64u ? sc.sobolBytes[g * 1024u + (k & 1023u)] : 0u;
        }
        __syncthreads();
    }
    const uint32_t count      = countIn ? *countIn : f.npix;
    __shared__ uint32_t pushLds[2][2][2 * PRT_SHADE_NW];  // [trip parity][shadow | next]: a trip's slots are read while the next trip counts
    const bool     literal    = (flags & PRT_FLAG_LITERAL) != 0;
    const bool     rr         = (flags & PRT_FLAG_RUSSIAN_ROULETTE) != 0;
    const int      maxBounces = ubo->numIndirectBounces;
    const int      numAA      = ubo->numAArays;

#ifdef PRT_SHADE_STAMPS  // measurement builds only: where a trip's cycles go (sums over waves, lane 0) -> stats[36..41]
    unsigned long long stampT[6] = {0, 0, 0, 0, 0, 0};
#define PRT_STAMP(k)                                                     \
    {                                                                    \
        __builtin_amdgcn_s_waitcnt(0);                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        stampT[k] += now_ - stampMark;                                   \
        stampMark = now_;                                                \
    }
#else
#define PRT_STAMP(k)
#endif
    // The loop is software-pipelined by hand: the path state of the NEXT trip is requested before this trip's results
    // are stored. vmcnt counts loads and stores in one in-order queue, so a load that is issued after the stores cannot
    // be waited for without waiting for every store's acknowledgement as well (thousands of cycles for the scattered
    // ones); requested before them, the state is there when the next trip starts.
    struct SlotState
    {
        uint32_t px;
        float4   hit, rayD, thr;
        bool     dead;
    };
    auto load_state = [&](uint32_t i) {
        SlotState st;
        st.px   = 0;
        st.hit  = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        st.rayD = st.hit;
        st.thr  = st.hit;
        st.dead = false;
        if(i < count)
        {
            st.px   = bounce ? ld_nt(gin.px + i) : i;
            st.hit  = ld_nt(gin.hit + i);
            st.rayD = ld_nt(gin.rayD + i);
            st.thr  = bounce ? ld_nt(gin.thr + i) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // primary rays: k_generate
            st.dead = (bounce == 0) && f.dead[i];  // bounce 0: slot == pixel
        }
        return st;
    };
    SlotState nx     = load_state(blockIdx.x * PRT_SHADE_BLOCK + threadIdx.x);
    uint32_t  parity = 0;
    for(uint32_t base = blockIdx.x * PRT_SHADE_BLOCK; base < count; base += gridDim.x * PRT_SHADE_BLOCK, parity ^= 1u)  // block-uniform trip count
    {
#ifdef PRT_SHADE_STAMPS
        unsigned long long stampMark = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t  i       = base + threadIdx.x;
        const SlotState in      = nx;
        const uint32_t  px      = in.px;
        const float4    hq0     = in.hit, rd4 = in.rayD, th4 = in.thr;
        bool            nextQ   = false, endQ = false;
        // every vertex that is shaded sends a shadow ray: its queue slot is reserved now, and the atomic's round trip
        // runs beside the shading
        const bool shadowQ = i < count && (uint32_t)__float_as_int(hq0.x) != 0xffffffffu && !in.dead;
        // what the vertex leaves behind is stored after the compactions, at the slots they hand out
        float4 oOrg = hq0, oShD = hq0, oEiV = hq0, oEiO = hq0, oDir = hq0, oThr = hq0;
        const PushTicket tShadow = block_push_begin(shadowQ, countShadow, pushLds[parity][0]);
        PRT_STAMP(0)  // path state of the slot, the shadow queue's ticket
        if(i < count)
        {
            const float4   hq  = hq0;
            // nvert[px] is written once per path, where the path ends: at the vertex that is not continued, or by the
            // miss that follows the last vertex. Bit 0x80 (set by k_generate in literal mode only) is kept.
            const uint32_t primitiveID = (uint32_t)__float_as_int(hq.x);
            const bool     isDead      = in.dead;
            if(primitiveID == 0xffffffffu || isDead)
            {
                if(bounce == 0 && !isDead)
                {
                    // primary miss — pathRT.rgen:693-697
                    const uint32_t ip = px % f.npixImg;
                    const uint32_t x = ip % f.w, y = ip / f.w;
                    const float    u = ((float)x + 0.5f) / (float)f.w, v = ((float)y + 0.5f) / (float)f.h;
                    // literal: the miss colour replaces E; default: it is one more term of the sum
                    f.contrib[px] = make_float4(u, v, 0.4f, 1.0f);
                    f.nvert[px]   = (uint8_t)((literal ? (f.nvert[px] & 0x80u) | 0x41u : 0x01u));
                    f.dead[px]    = 1;
                }
                else if(bounce != 0)
                    endQ = true;  // `bounce` terms stored
            }
            else
            {
                const f3     Rd  = mk3(rd4.x, rd4.y, rd4.z);
                f3           throughput = mk3(th4.x, th4.y, th4.z);
                float        p          = th4.w;
                const f3     bary       = mk3(1.0f - hq.y - hq.z, hq.y, hq.z);  // pathRT.rchit:19

                // sobolIndex = layer0 / 2 + iteration, +1 per finished aaRay (:627-631, :970); the aaRay
                // ordinal travels in bits 8..15 of `flags`.
                const uint32_t sub  = px / f.npixImg, ip = px - sub * f.npixImg;  // iteration of the batch, image pixel
                const uint32_t s0   = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 0);
                const uint32_t sidx = s0 / 2u + (ubo->iteration + sub * f.iterStride) + ((flags >> 8) & 0xffu);
                const uint32_t layer = 1u + bounce;  // scrambleArray[scrambleArrayLayer++], :664,:703
                const uint32_t scr0  = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 2u * layer);
                const uint32_t scr1  = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 2u * layer + 1u);
                // sobolDim after the pixel offset is 4 + 4*bounce, 4 per vertex: tables 0..3 of sobT

                const TriU     tri = TWO ? load_tri_inst(sc, primitiveID) : load_tri(sc.stri, primitiveID);
                const VertexU &v0 = tri.v0, &v1 = tri.v1, &v2 = tri.v2;
                const uint32_t      mi  = (tri.mat >= 0 && (uint32_t)tri.mat < sc.nmats) ? (uint32_t)tri.mat : 0u;
                const prt_material& mat = sc.mats[mi];

                PRT_STAMP(1)  // shading triangle (+ whatever else the compiler hoisted)
                f3 sNormal = normalize3((v0.normal * bary.x + v1.normal * bary.y) + v2.normal * bary.z);
                if(dot3(sNormal, Rd) > 0.0f) sNormal = -sNormal;
                const f3 gNormal  = sNormal;
                const f3 hitPoint = (v0.pos * bary.x + v1.pos * bary.y) + v2.pos * bary.z;

                f3 albedo   = mk3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
                f3 specular = mk3(mat.specular[0], mat.specular[1], mat.specular[2]);
                f3 textureN = mk3(1.0f);
                f2 texCoord;
                texCoord.x = (v0.uv.x * bary.x + v1.uv.x * bary.y) + v2.uv.x * bary.z;
                texCoord.y = (v0.uv.y * bary.x + v1.uv.y * bary.y) + v2.uv.y * bary.z;
                const int ntex = (int)sc.ntex;
                const int tD = mat.diffuseTextureID, tS = mat.specularTextureID, tN = mat.normalTextureID;
                if(tD >= 0 && tD < ntex) albedo = albedo * texture_rgb(sc.tex[tD], texCoord);
                if(tS >= 0 && tS < ntex) specular = specular * texture_rgb(sc.tex[tS], texCoord);
                if(tN >= 0 && tN < ntex)
                {
                    textureN = texture_rgb(sc.tex[tN], texCoord) * 2.0f - mk3(1.0f);
                    // bump mapping — :754-783
                    const f3    v10 = v1.pos - v0.pos, v20 = v2.pos - v0.pos;
                    const float u10x = v1.uv.x - v0.uv.x, u10y = v1.uv.y - v0.uv.y;
                    const float u20x = v2.uv.x - v0.uv.x, u20y = v2.uv.y - v0.uv.y;
                    const float det  = u10x * u20y - u20x * u10y;
                    if(gabs(det) > 1e-4f)
                    {
                        const float od    = 1.0f / det;
                        const float c0x = u20y * od, c0y = -u10y * od;
                        const float c1x = -u20x * od, c1y = u10x * od;
                        m3 nm;
                        nm.c0   = normalize3(v10 * c0x + v20 * c0y);
                        nm.c1   = normalize3(v10 * c1x + v20 * c1y);
                        nm.c2   = sNormal;
                        sNormal = mulM(nm, textureN);
                    }
                }
                {
                    const float mx = maxcoord(albedo + specular);  // :786-793
                    if(mx > 1.0f)
                    {
                        albedo   = albedo / mk3(mx);
                        specular = specular / mk3(mx);
                    }
                }

                // light sample — :798-805, sampleLight :304-309
                f2 s;
                s.x = sobol1D_bytes(sobT, sidx, scr0);
                s.y = sobol1D_bytes(sobT + 1024, sidx, scr1);
                float       pdf = 1.0f / (4.0f * ubo->lightSize[0] * ubo->lightSize[1]);
                const float lpx = (s.x * 2.0f - 1.0f) * ubo->lightSize[0];
                const float lpy = (s.y * 2.0f - 1.0f) * ubo->lightSize[1];
                float       lp[4];
                mul44(ubo->lightTransform, lpx, lpy, 0.0f, 1.0f, lp);
                const f3 vLight = mk3(lp[0], lp[1], lp[2]) - hitPoint;

                const float roughness = gclamp(1.0f - mat.shininess, 0.001f, 1.0f);
                const float alpha     = roughness * roughness;
                const m3    l2w       = form_basis(gNormal);
                const f3    wo        = normalize3(mulMT(l2w, -Rd));

                // direct term, evaluated as if the light sample were visible — :824-847
                f3 EiV;
                {
                    const f3    lightNormal = -normalize3(mk3(ubo->lightTransform[8], ubo->lightTransform[9],
                                                               ubo->lightTransform[10]));
                    const float r           = length3(vLight);
                    const float cosL        = gclamp(dot3(normalize3(-vLight), lightNormal), 0.0f, 1.0f);
                    const float cosS        = gclamp(dot3(sNormal, normalize3(vLight)), 0.0f, 1.0f);
                    const f3    wiL         = normalize3(mulMT(l2w, vLight));
                    const f3    wmL         = normalize3(wo + vLight);  // sic (:837)
                    const f3    F           = ggx_schlick(specular, dot3(wiL, wmL));
                    const float G1          = ggx_smith_masking(wo, alpha);
                    const float G2          = ggx_smith_masking_shadowing(wiL, wo, alpha);
                    const f3    brdf        = F * (G2 / G1);
                    const f3    lightE      = mk3(ubo->lightE[0], ubo->lightE[1], ubo->lightE[2]);
                    EiV = mk3(0.0f) + lightE * cosL * cosS / (r * r * pdf);
                    EiV = EiV * (brdf * throughput);
                }

                // BSDF sample — :849-918
                f3 wi           = mk3(0.0f);
                f3 specularBRDF = mk3(0.0f);
                {
                    f2 rnd;
                    rnd.x       = sobol1D_bytes(sobT + 2048, sidx, scr0);
                    rnd.y       = sobol1D_bytes(sobT + 3072, sidx, scr1);
                    const f3 wm = ggx_sample_vndf(wo, alpha, rnd);
                    const float fres = ggx_fresnel_dielectric(wo.z, 1.0f, 1.45f);
                    if(mat.dissolve == 0.0f)
                    {
                        if(rnd.x < fres)
                        {
                            wi = normalize3(reflect3(-wo, wm));
                        }
                        else
                        {
                            if(!refract3(wo, mk3(0.0f, 0.0f, 1.0f), wi, 1.0f / 1.45f)) wi = reflect3(-wo, wm);
                            wi = normalize3(wi);
                        }
                    }
                    else
                    {
                        wi = normalize3(reflect3(-wo, wm));
                    }
                    if(wi.z > 0.0f)
                    {
                        const float D   = ggx_distribution(wm, alpha);
                        const float G1v = ggx_g1(wo, wm, alpha);
                        if(mat.dissolve == 1.0f)
                        {
                            const f3    F  = ggx_schlick(specular, dot3(wi, wm));
                            const float G1 = gclamp(ggx_smith_masking(wo, alpha), 1e-4f, 1.0f);
                            const float G2 = ggx_smith_masking_shadowing(wi, wo, alpha);
                            specular       = F * G1v;
                            specularBRDF   = F * (G2 / G1);
                        }
                        else
                        {
                            pdf          = 1.0f - gclamp(fres, 0.0f, 1.0f - 1e-2f);
                            specularBRDF = albedo / pdf;
                        }
                        p = G1v * gabs(dot3(wm, wi)) * D / wi.z;
                        p *= 1.0f / (4.0f * gabs(dot3(wo, wm)));
                    }
                }
                wi          = mulM(l2w, wi);
                const f3 Rn = wi * mk3(1000.0f);

                f3    color  = albedo * 1.0f + specular * specularBRDF;
                float colorw = 1.0f;
                if(numAA != 1)
                {
                    const float2 ro     = f.aaOff[px];
                    const float  weight = mitchell_netrevali(ro.x + 0.5f) * mitchell_netrevali(ro.y + 0.5f);
                    color               = mk3(color.x / weight, color.y / weight, color.z / weight);
                    colorw              = colorw / weight;
                }

                PRT_STAMP(2)  // material, textures, light sample, BSDF sample
                // accumulate — :939-942; the shadow kernel picks one of the two, k_resolve adds it to E
                const f3 em  = mk3(mat.emission[0], mat.emission[1], mat.emission[2]) * ubo->lightOtherE;
                const f3 tc  = throughput * color;
                f3       EiO = mk3(0.0f) + em;
                EiO          = EiO * tc;
                EiV          = EiV + em;
                EiV          = EiV * tc;
                // origin of the shadow ray and of the extension ray alike (:805, :922), tmin 1e-6 for both
                oOrg = make_float4(hitPoint.x, hitPoint.y, hitPoint.z, 0.000001f);
                oShD = make_float4(vLight.x, vLight.y, vLight.z, 0.0f);
                oEiV = make_float4(EiV.x, EiV.y, EiV.z, colorw);
                oEiO = make_float4(EiO.x, EiO.y, EiO.z, colorw);

                // termination — :944-961
                bool alive = true;
                if(p != p) alive = false;  // isnan(p)
                if(alive && p == 0.0f) alive = false;
                if(alive)
                {
                    throughput = throughput * color;
                    if((int)(bounce + 1u) > maxBounces) alive = false;
                }
                if(alive && rr && (bounce + 1u) >= 2u)
                {
                    // Russian roulette (extension, off by default)
                    const uint32_t b1  = bounce + 1u;
                    const uint32_t rs  = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 30u + (b1 & 1u));
                    const float    q   = gclamp(maxcoord(throughput), 0.05f, 1.0f);
                    const float    xi  = sobol1D_bytes(sobT + 4096, sidx, rs);
                    if(xi >= q)
                        alive = false;
                    else
                        throughput = throughput / mk3(q);
                }
                if(!alive)
                    endQ = true;
                else
                {
                    oDir  = make_float4(Rn.x, Rn.y, Rn.z, 1.0f);
                    oThr  = make_float4(throughput.x, throughput.y, throughput.z, p);
                    nextQ = true;
                }
            }
        }
        PRT_STAMP(3)  // stores, termination
        const PushTicket tNext = block_push_begin(nextQ, countNext, pushLds[parity][1]);
        nx = load_state(i + gridDim.x * PRT_SHADE_BLOCK);
        block_push_publish(tShadow, pushLds[parity][0]);
        block_push_publish(tNext, pushLds[parity][1]);
        __syncthreads();
        PRT_STAMP(4)  // compaction
        if(endQ)  // the path ends here: terms stored = vertices shaded (this one included when it was a hit)
            f.nvert[px] = (uint8_t)((literal ? (uint32_t)(f.nvert[px] & 0x80u) : 0u) | (bounce + (shadowQ ? 1u : 0u)));
        if(shadowQ)
        {
            const uint32_t s = block_push_slot(tShadow, pushLds[parity][0]);
            st_nt(f.contrib + ((size_t)bounce * f.npix + px), oEiO);
            st_nt(f.shPx + s, px);
            st_nt(f.shO + s, oOrg);
            st_nt(f.shD + s, oShD);
            st_nt(f.eiVis + s, oEiV);
        }
        if(nextQ)
        {
            const uint32_t s = block_push_slot(tNext, pushLds[parity][1]);
            st_nt(gout.px + s, px);
            st_nt(gout.rayO + s, oOrg);
            st_nt(gout.rayD + s, oDir);
            st_nt(gout.thr + s, oThr);
        }
        PRT_STAMP(5)  // stores by slot
    }
#ifdef PRT_SHADE_STAMPS
    if((threadIdx.x & 63) == 0)
        for(int k = 0; k < 6; ++k) atomicAdd(f.stats + 36 + k, stampT[k]);
    if(threadIdx.x == 0) atomicAdd(f.stats + 42, (unsigned long long)((count + PRT_SHADE_BLOCK - 1u) / PRT_SHADE_BLOCK > blockIdx.x ? ((count + PRT_SHADE_BLOCK - 1u) / PRT_SHADE_BLOCK - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u));
#endif
#undef PRT_STAMP
}

// ---------------------------------------------------------------------------------------------
// shadow + connect — traceNV(terminate on first hit) :818-822, then the accumulation of :939-942
// ---------------------------------------------------------------------------------------------
struct ShadowIO
{
    const FrameView& f;
    float4*          term;  // contrib + bounce * npix
    PRT_DEV bool load(uint32_t i, f3& o, float& tmin, f3& d, float& tmax, uint32_t& tag) const
    {
        const float4 oq = ld_nt(f.shO + i), dq = ld_nt(f.shD + i);
        o    = mk3(oq.x, oq.y, oq.z);
        tmin = 0.000001f;
        d    = mk3(dq.x, dq.y, dq.z);
        tmax = 1.0f;
        tag  = i;
        return true;
    }
    PRT_DEV void store(uint32_t i, const Hit& h) const
    {
        if(h.prim == 0xffffffffu) st_res(term + ld_nt(f.shPx + i), ld_nt(f.eiVis + i));  // visible: Ei_occ -> Ei_vis
    }
};

template <bool INSTR, bool TWO = false>
__global__ void __launch_bounds__(PRT_TRACE_BLOCK, INSTR ? 4 : (TWO ? PRT_TRACE_WAVES_TWO : PRT_TRACE_WAVES)) PRT_TRACE_ATTR k_shadow(SceneView sc, FrameView f,
                                                            const uint32_t* __restrict__ countPtr, uint32_t* workCtr,
                                                            uint32_t bounce, TraceTune tune)
{
    __shared__ __align__(16) uint32_t stackLds[TWO ? PRT_TRACE_LDS_WORDS_TWO : PRT_TRACE_LDS_WORDS];
    const uint4* topLds = stage_top_nodes<PRT_LS(TWO)>(sc.bvh, stackLds);
    const uint32_t      count = *countPtr;
    uint32_t            nskip = 0;
    TraceCount          tc    = {};
    ShadowIO            io{f, f.contrib + (size_t)bounce * f.npix};
    trace_stream<true, INSTR, ShadowIO, TWO>(sc.bvh, &stackLds[threadIdx.x], topLds, count, workCtr, tune, io, nskip, tc);
    flush_counts<INSTR>(f, count, nskip, tc, 1);
}

// The shadow rays of bounce b and the extension rays of bounce b+1 both come out of k_shade(b) and do not depend on
// each other: one launch traces both. A wave first drains the extension queue (the next k_shade waits for it), then
// the shadow queue, so the few long rays at the end of the first stream run beside the second stream instead of on
// an empty GPU — one launch tail per bounce instead of two.
template <bool INSTR, bool TWO = false>
__global__ void __launch_bounds__(PRT_TRACE_BLOCK, INSTR ? 4 : (TWO ? PRT_TRACE_WAVES_TWO : PRT_TRACE_WAVES)) PRT_TRACE_ATTR k_shadow_extend(
    SceneView sc, FrameView f, const uint32_t* __restrict__ shCountPtr, uint32_t* shWork, uint32_t bounce,
    const uint32_t* __restrict__ exCountPtr, uint32_t* exWork, TraceTune tune)
{
    __shared__ __align__(16) uint32_t stackLds[TWO ? PRT_TRACE_LDS_WORDS_TWO : PRT_TRACE_LDS_WORDS];
    const uint4* topLds = stage_top_nodes<PRT_LS(TWO)>(sc.bvh, stackLds);
    {
        const uint32_t count = *exCountPtr;
        uint32_t       nskip = 0;
        TraceCount     tc    = {};
        ExtendIO       io(f.gen[(bounce + 1u) & 1u], nullptr);
        trace_stream<false, INSTR, ExtendIO, TWO>(sc.bvh, &stackLds[threadIdx.x], topLds, count, exWork, tune, io, nskip, tc);
        flush_counts<INSTR>(f, count, nskip, tc, 0);
        if(blockIdx.x == 0 && threadIdx.x == 0 && count) atomicAdd(f.stats + 6, (unsigned long long)count);
    }
    {
        const uint32_t count = *shCountPtr;
        uint32_t       nskip = 0;
        TraceCount     tc    = {};
        ShadowIO       io{f, f.contrib + (size_t)bounce * f.npix};
        trace_stream<true, INSTR, ShadowIO, TWO>(sc.bvh, &stackLds[threadIdx.x], topLds, count, shWork, tune, io, nskip, tc);
        flush_counts<INSTR>(f, count, nskip, tc, 1);
        if(blockIdx.x == 0 && threadIdx.x == 0 && count) atomicAdd(f.stats + 7, (unsigned long long)count);
    }
}

// ---------------------------------------------------------------------------------------------
// resolve — E += Ei of every path vertex, in path order (pathRT.rgen:939-942, :649-654, :693-697, :973)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_resolve(float4* __restrict__ accum, const float4* __restrict__ contrib,
                                                 const uint8_t* __restrict__ nvert, uint32_t npixImg, uint32_t nsub)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x;
    if(px >= npixImg) return;
    const size_t npix = (size_t)npixImg * nsub;  // pixels of the pass = row length of contrib
    float4       E    = accum[px];
    bool         any  = false;
    for(uint32_t s = 0; s < nsub; ++s)  // the iterations of the batch, in iteration order
    {
        const size_t   vp = (size_t)s * npixImg + px;
        const uint32_t n  = nvert[vp];
        if(n == 0) continue;
        any = true;
        if(n & 0x80u) E = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if(n & 0x40u)
            E = ld_nt(contrib + vp);
        else
        {
            const uint32_t cnt = n & 0x3fu;
            for(uint32_t b = 0; b < cnt; ++b)
            {
                const float4 t = ld_nt(contrib + (size_t)b * npix + vp);
                E              = make_float4(E.x + t.x, E.y + t.y, E.z + t.z, E.w + t.w);
            }
        }
    }
    if(any) accum[px] = E;
}

// ---------------------------------------------------------------------------------------------
// AO — AO.rgen:214-272, after k_ao_generate + k_extend
// ---------------------------------------------------------------------------------------------
// BATCH: the pass holds several iterations side by side (prt_render); the iteration's term goes to contrib[px] and
// k_resolve adds the terms of a pixel to E in iteration order, which is the order E.x + ao is evaluated in below.
template <bool TWO = false, bool BATCH = false>
__global__ void __launch_bounds__(PRT_TRACE_BLOCK) k_ao_shade(SceneView sc, FrameView f, const prt_uniforms ubo_,
                                                              uint32_t flags, uint32_t* workCtr)
{
    const prt_uniforms* ubo = &ubo_;
    __shared__ __align__(16) uint32_t stackLds[TWO ? PRT_TRACE_LDS_WORDS_TWO : PRT_TRACE_LDS_WORDS];
    // Sobol byte tables of dimensions 0 and 1 (sobol1D_bytes: four reads instead of the 32-step loop, same bits)
    __shared__ uint32_t sobT[2 * 1024];
    for(uint32_t k = threadIdx.x; k < 2u * 1024u; k += PRT_TRACE_BLOCK) sobT[k] = sc.sobolBytes[k];
    const uint4* topLds = stage_top_nodes<PRT_LS(TWO)>(sc.bvh, stackLds);  // (its barrier covers sobT as well)
    __syncthreads();
    const uint32_t      count   = f.npix;
    const int           lane    = threadIdx.x & 63;
    const bool          literal = (flags & PRT_FLAG_LITERAL) != 0;
    const int           numAO   = ubo->numAOrays;
    const float         aoLen   = ubo->aoRayLength;
    uint32_t            nrays   = 0;
    // 256 pixels per work-counter atomic (a device-scope atomic is a round trip to the memory side), 64 at a time
    for(uint32_t sub = 4u, base0 = 0u;; ++sub)
    {
        if(sub == 4u)
        {
            base0 = wave_fetch(workCtr, 256u);
            if(base0 >= count) break;
            sub = 0u;
        }
        const uint32_t px = base0 + sub * 64u + lane;
        if(px < count)
        {
            const float4   hq   = f.gen[0].hit[px];
            const uint32_t prim = (uint32_t)__float_as_int(hq.x);
            const uint32_t ip   = BATCH ? px % f.npixImg : px;
            float4         E    = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if(!BATCH) E = f.accum[px];
            if(BATCH) f.nvert[px] = 1;  // one term per iteration
            if(prim == 0xffffffffu)
            {
                if(BATCH)
                    st_nt(f.contrib + px, make_float4(0.1f, 0.1f, 0.1f, 0.0f));
                else
                    f.accum[px] = literal ? make_float4(0.1f, 0.1f, 0.1f, 0.0f)
                                          : make_float4(E.x + 0.1f, E.y + 0.1f, E.z + 0.1f, E.w + 0.0f);
            }
            else
            {
                const uint32_t scr0 = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 0);
                const uint32_t scr1 = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 1);
                uint32_t sobolIndex = scramble_value(sc.scrTable, f.npixImg, sc.seed, ip, 2);
                const TriU     tri = TWO ? load_tri_inst(sc, prim) : load_tri(sc.stri, prim);
                const VertexU &v0 = tri.v0, &v1 = tri.v1, &v2 = tri.v2;
                const f3      bary = mk3(1.0f - hq.y - hq.z, hq.y, hq.z);
                const float4  rd4  = f.gen[0].rayD[px];
                f3 normal = normalize3((v0.normal * bary.x + v1.normal * bary.y) + v2.normal * bary.z);
                if(dot3(normal, mk3(rd4.x, rd4.y, rd4.z)) > 0.0f) normal = normal * -1.0f;
                const m3 onb      = form_basis(normal);
                f3       hitPoint = (v0.pos * bary.x + v1.pos * bary.y) + v2.pos * bary.z;
                hitPoint          = hitPoint + 0.0001f * normal;
                int noHit = 0;
                for(int k = 0; k < numAO; ++k)
                {
                    // hemisphereSample — AO.rgen:153-181
                    const float s0 = sobol1D_bytes(sobT, sobolIndex, scr0);
                    const float s1 = sobol1D_bytes(sobT + 1024, sobolIndex, scr1);
                    sobolIndex++;
                    float       phi, r;
                    const float a = 2.0f * s0 - 1.0f;
                    const float b = 2.0f * s1 - 1.0f;
                    if(a * a > b * b)
                    {
                        r   = a;
                        phi = kPi * 0.25f * (b / a);
                    }
                    else
                    {
                        r   = b;
                        phi = kPi * 0.5f - kPi * 0.25f * (a / b);
                    }
                    float sn, cs;
                    sincos_fixed(phi, sn, cs);
                    const float hx = r * cs;
                    const float hy = r * sn;
                    const float hz = __builtin_sqrtf(gmax(0.0f, 1.0f - hx * hx - hy * hy));
                    const f3    Rd = normalize3(mulM(onb, mk3(hx, hy, hz))) * mk3(aoLen);
                    const Hit   h  = trace_ray<true, false, TWO>(sc.bvh, &stackLds[threadIdx.x], topLds, hitPoint, 0.00001f, Rd, 1.0f);
                    if(h.prim == 0xffffffffu) noHit++;
                }
                nrays += (uint32_t)numAO;
                const float ao = (float)noHit / (float)numAO;
                if(BATCH)
                    st_nt(f.contrib + px, make_float4(ao, ao, ao, 1.0f));
                else
                    f.accum[px] = make_float4(E.x + ao, E.y + ao, E.z + ao, E.w + 1.0f);
            }
        }
    }
    {
        const uint32_t r = wave_sum(nrays);
        if((threadIdx.x & 63) == 0 && r) atomicAdd(f.stats + 1, (unsigned long long)r);
    }
}

// pathRTpostprocess.comp:38-51 over the whole image (the reference's W/16 x H/16 dispatch drops
// edge pixels when W,H are not multiples of 16, vkRTX_setup.cpp:779).
__global__ void __launch_bounds__(256) k_postprocess(const float4* __restrict__ accum, uint32_t npix,
                                                     uint32_t* __restrict__ rgba8)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= npix) return;
    float4 c = accum[i];
    if(c.w != 0.0f)
    {
        c.x /= c.w;
        c.y /= c.w;
        c.z /= c.w;
        c.w /= c.w;
    }
    c.x = gamma_fixed(c.x);
    c.y = gamma_fixed(c.y);
    c.z = gamma_fixed(c.z);
    const float v[4] = {c.x, c.y, c.z, c.w};
    uint32_t    out  = 0;
    for(int k = 0; k < 4; ++k)
    {
        float t = v[k];
        t       = !(t > 0.0f) ? 0.0f : (t > 1.0f ? 1.0f : t);
        out |= (uint32_t)(t * 255.0f + 0.5f) << (8 * k);
    }
    rgba8[i] = out;
}

// generic batch trace for prt_trace() (unit tests / ray-level benchmarks): same persistent stream
struct BatchIO
{
    const float4* rays;
    float4*       hits;
    PRT_DEV bool load(uint32_t i, f3& o, float& tmin, f3& d, float& tmax, uint32_t& tag) const
    {
        const float4 oq = rays[2 * (size_t)i], dq = rays[2 * (size_t)i + 1];
        o    = mk3(oq.x, oq.y, oq.z);
        tmin = oq.w;
        d    = mk3(dq.x, dq.y, dq.z);
        tmax = dq.w;
        tag  = i;
        return true;
    }
    PRT_DEV void store(uint32_t i, const Hit& h) const
    {
        hits[i] = make_float4(__int_as_float((int)h.prim), h.u, h.v, h.t);
    }
};
template <bool ANY, bool TWO = false>
__global__ void __launch_bounds__(PRT_TRACE_BLOCK, TWO ? PRT_TRACE_WAVES_TWO : PRT_TRACE_WAVES) PRT_TRACE_ATTR k_trace_batch(BvhView bvh, const float4* __restrict__ rays,
                                                                 uint32_t n, float4* __restrict__ hits,
                                                                 uint32_t* workCtr, TraceTune tune,
                                                                 unsigned long long* stats)
{
    __shared__ __align__(16) uint32_t stackLds[TWO ? PRT_TRACE_LDS_WORDS_TWO : PRT_TRACE_LDS_WORDS];
    const uint4* topLds = stage_top_nodes<PRT_LS(TWO)>(bvh, stackLds);
    uint32_t            nskip = 0;
    TraceCount          tc    = {};
    BatchIO             io{rays, hits};
    trace_stream<ANY, false, BatchIO, TWO>(bvh, &stackLds[threadIdx.x], topLds, n, workCtr, tune, io, nskip, tc);
    if(tc.overrun && (threadIdx.x & 63) == 0) atomicAdd(stats + 28, 1ull);
}

}  // namespace prt
