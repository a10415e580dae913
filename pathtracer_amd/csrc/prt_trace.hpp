// prt_trace.hpp — wave64 BVH traversal for gfx950: replaces traceNV / the RT cores
// (shaders/pathRT.rgen:692,819-822,963; shaders/AO.rgen:215,259-262).
//
// * BVH4, 64-byte nodes (two per HBM/L2 line), four 16-byte loads per visit: the child boxes are 16-bit
//   fixed point on the scene's bounding box (rounded outwards; one row per axis: min[4] | max[4]) + one
//   row of four child references. The traversal kernels are bound by the vector-memory pipeline
//   (TA/TCP: random 16-byte requests), not by VALU or HBM, so bytes and loads per visit are what counts;
//   the de-quantisation folds into the per-ray slab constants: t = fma(float(q), r*scale, c').
// * Leaves reference runs of <= 2 pre-gathered triangles in 64-byte slots (one sector: what a divergent gather costs
//   is the number of sectors it touches, tools/ubench_gather.hip): a single triangle is 3 x float4 (prim id in v0.w), two
//   triangles that form a fan (A, B, C), (A, C, D) — 9 leaves in 10 — are their four corners in ONE slot, any other pair
//   takes two slots (leaf_test).
// * Per-lane traversal stack: the first PRT_LDS_STACK entries live in LDS, laid out [entry][lane] so
//   that a wave's push/pop is one conflict-free ds_write_b32/ds_read_b32; deeper entries (rare)
//   spill to scratch. The same LDS words hold the part of a ray's state that the node steps never touch (tag, best hit's
//   prim / u / v): with 9 slab constants per ray that is what lets the kernels run at 7 waves per SIMD (72 VGPRs).
// * trace_stream(): persistent waves. A wave keeps up to 64 rays in flight; lanes whose ray has
//   finished are refilled from the queue (chunked work fetch, one atomic per PRT_CHUNK rays), node
//   steps and triangle tests run in separate wave-uniform phases (a found leaf is postponed while the
//   lane keeps descending), so that divergence between "at a node" and "at a leaf" lanes and between
//   short and long rays does not idle the SIMD.
// * Watertight ray/triangle test (Woop, Benthin, Wald 2013) with the fp64 fallback for edge
//   functions that evaluate to exactly 0; two-sided triangles (TRIANGLE_CULL_DISABLE,
//   vkRT_TLAS.cpp:218-223); hit iff tmin < t < tmax; closest hit ties -> lower primitive id.
//   The result of a trace is therefore independent of the tree and of the traversal order; the box
//   test only has to be conservative: boxes grown by 64*2^-24*max(|origin|, |scene|), which covers v_rcp_f32, the
//   roundings of the slab arithmetic and the tolerance of the triangle test (budget in ray_prepare).
// * Two-level scenes (instances of bottom-level structures, vkRT_TLAS.cpp:58-72): the TWO instantiations of
//   trace_ray / trace_stream walk a top-level BVH4 over the instances' world boxes in the same node array; a lane
//   that reaches an instance takes its ray to object space (same t), pushes an exit marker and walks the
//   instance's structure; popping the marker restores the world-space ray. A hit is defined in object space
//   (oracle.cpp: "instances"), so the top level, like every box test, only has to be conservative.
#pragma once
#include "prt_device.hpp"

namespace prt {

#ifndef PRT_LDS_STACK
#define PRT_LDS_STACK 12  // entries per lane in LDS (deeper ones: scratch); 12 + PRT_COLD_WORDS = 16 KiB per block of 256
#endif
#ifndef PRT_LDS_STACK_TWO
#define PRT_LDS_STACK_TWO 16  // two-level kernels (6 waves per SIMD: 6 x (20 + 6) KiB; top-level entries + marker + the structure's)
#endif
// Loading a leaf's next triangle while the current one is tested costs 14 VGPRs = the sixth wave per SIMD. With
// leaves of <= 2 triangles the sixth wave is worth more (measured: 2.33 -> 2.30 ms); with leaves of <= 4 it was not.
#ifndef PRT_TRI_PREFETCH
#define PRT_TRI_PREFETCH 0
#endif
#ifndef PRT_NODE_ASM
#define PRT_NODE_ASM 0
#endif
// leaf records: 64-byte slots; two triangles that form a fan share one slot (leaf_test)
#define PRT_LEAF_SLOT_F4 4
#ifndef PRT_LEAF_QUAD
#if defined(PRT_LEAF_MAX) && PRT_LEAF_MAX != 2
#define PRT_LEAF_QUAD 0  // code 3 means three triangles there
#else
#define PRT_LEAF_QUAD 1
#endif
#endif
// A pair of triangles that is no fan takes two slots. 1: a leaf test handles the first one and hands back a reference to
// the second (a single-triangle leaf in the next slot), which the lane keeps as its postponed leaf for a later triangle
// round — no dependent second fetch inside the round that every lane of the wave would wait for.
#ifndef PRT_LEAF_SPLIT_PAIR
#if PRT_LEAF_QUAD
#define PRT_LEAF_SPLIT_PAIR 1  // k_shadow_extend 12.85 -> 12.55 ms per launch (profiles/r03_exp_split_pair.log)
#else
#define PRT_LEAF_SPLIT_PAIR 0
#endif
#endif
#if PRT_LEAF_SPLIT_PAIR && !PRT_LEAF_QUAD
#error "PRT_LEAF_SPLIT_PAIR needs the one-slot leaf forms of PRT_LEAF_QUAD"
#endif
#define PRT_SPILL_STACK (100 - PRT_LDS_STACK)
#define PRT_MAX_DEPTH 64  // depth limit of the binary build tree (=> <= 33 BVH4 levels, <= 100 stack entries)
#ifndef PRT_TRACE_BLOCK
#define PRT_TRACE_BLOCK 256
#endif
#ifndef PRT_TRACE_WAVES
// waves per SIMD the traversal kernels are compiled for: 7 = 72 VGPRs (9 slab constants instead of 12, the cold part of the
// ray state in LDS) and 7 x 22 KiB of LDS per CU. Round 3, same bench: 6 waves / 20 entries 13.67 ms per launch of
// k_shadow_extend, 6 / 16 with the register diet 13.26, 7 / 12: 12.83.
#define PRT_TRACE_WAVES 7
#endif
#define PRT_NODE_BYTES 64
// The first PRT_TOP_NODES BVH4 nodes (numbered level by level: the top of the tree) are copied into LDS by every block.
// More than half of all node visits go there (bench scene: 7.0 of 13.0 per closest-hit ray, 8.4 of 14.5 per any-hit ray in
// the first 96 nodes), and what the L1 charges for a divergent load is per LANE, hit or miss (tools/ubench_gather.hip
// modes G / H: lanes that read a hot 85-record set through the L1 cost as much as lanes that miss; from LDS they cost
// nothing there) — the L1's lane rate is what bounds the node phases.
#ifndef PRT_TOP_NODES
#define PRT_TOP_NODES 96  // 6 KiB: 7 blocks x (16 + 6) KiB = 154 of the CU's 160 KiB
#endif
// Per-lane LDS words of the traversal kernels: the stack entries, then the part of a ray's state that the node steps never
// touch (trace_stream: the ray's tag and the best hit's prim / u / v, written when a closer hit is found and read when the
// ray is finished) — four registers fewer in the loop that decides the occupancy.
#define PRT_COLD_WORDS 4
#define PRT_TRACE_LDS_WORDS_OF(ls) (((ls) + PRT_COLD_WORDS) * PRT_TRACE_BLOCK + PRT_TOP_NODES * 16)
#define PRT_TRACE_LDS_WORDS PRT_TRACE_LDS_WORDS_OF(PRT_LDS_STACK)
#define PRT_TRACE_LDS_WORDS_TWO PRT_TRACE_LDS_WORDS_OF(PRT_LDS_STACK_TWO)
// LDS stack entries of an instantiation
#define PRT_LS(two) ((two) ? PRT_LDS_STACK_TWO : PRT_LDS_STACK)
#ifndef PRT_CHUNK
#define PRT_CHUNK 256u  // rays per work-counter atomic: a device-scope atomic is a round trip to the memory side (64: 7.6 ms per launch, 128: 4.32, 256: 4.20, 512: 4.24, 1024: 4.33; eight per-XCD cursors or a static share of the queue: no better)
#endif

static_assert(1 + 3 * (PRT_MAX_DEPTH / 2 + 1) <= PRT_LDS_STACK + PRT_SPILL_STACK && PRT_LDS_STACK >= 8 && PRT_LDS_STACK_TWO >= 8, "stack covers the deepest tree");

constexpr int32_t kSentinel  = 0x7fffffff;  // bottom-of-stack marker: the ray is finished
constexpr int32_t kInstExit  = 0x7ffffffe;  // two-level scenes: the entries above this one belong to an instance
constexpr int32_t kEmptyLeaf = -1;          // ~0: leaf reference with count 0
// two-level scenes: the top-level tree is limited to 32 binary levels (<= 51 stack entries), + 1 exit marker
#define PRT_TLAS_MAX_DEPTH 32
#define PRT_SPILL_STACK_TWO (100 - PRT_LDS_STACK_TWO + 3 * (PRT_TLAS_MAX_DEPTH / 2 + 1) + 1)

struct BvhView
{
    const uint4*  nodes;  // 4 x uint4 per BVH4 node
    const float4* tris;   // one 128-byte slot per triangle position; a leaf's triangles are packed into its first slot(s)
    int32_t       root;   // node index (0), or a leaf reference when the scene has <= 4 triangles
    float         sceneAbs;
    float         qmin[3], qscale[3];  // box coordinate = qmin + q * qscale, q in [0, 65535]
    // Two-level scenes (kernels instantiated with TWO): root / sceneAbs / the grid above are the top-level
    // structure's, whose leaves are single instances; all structures share `nodes` and `tris`.
    const float4* inst;      // 8 x float4 per instance, in top-level leaf order (InstanceRec below)
    float         padScale;  // widening of the top-level box test: covers the rounding of the object-space transforms
    uint32_t      nTop;      // nodes 0 .. nTop-1 are staged in LDS by the traversal kernels (<= PRT_TOP_NODES)
};
// Instance record, 128 bytes: rows 0-2 world -> object (row i = (W[0][i], W[1][i], W[2][i], W[3][i]) of the
// column-major inverse transform), row 3 = qmin.xyz | sceneAbs of the bottom-level structure, row 4 = qscale.xyz | root
// reference (int bits), row 5 = (flattened primitive id of its first triangle, flags: 1 = identity transform, -, -).
#define PRT_INST_F4 8

struct Hit
{
    uint32_t prim;
    float    u, v, t;
};

struct TraceCount
{
    uint32_t nodes, tris;
    // wave-level step counts (lane 0 only): how many lanes worked in each wave-wide node step / triangle step / ray
    // set-up, for the SIMD-utilisation figures of prt_get_profile
    uint32_t wNode, lNode, wTri, lTri, wSetup, lSetup;
    // time stamps (s_memtime ticks, lane 0): spent in ray set-up / node phases / triangle phases / whole stream
    uint32_t cSetup, cNode, cTri, cTotal;
    uint32_t overrun;  // the safety bound of trace_stream was reached: rays of this wave were dropped
};

// traversal knobs (wave-uniform kernel argument)
struct TraceTune
{
    uint32_t refillMin;  // refill when at least this many lanes are idle
    uint32_t stallMax;   // leave the node phase when more than this many active lanes wait at a leaf / are done
    uint32_t leafMin;    // run another triangle round only when at least this many lanes have a leaf again
    uint32_t pad;
};

struct RayCtx
{
    float    ox, oy, oz, tmin, tmax;
    float    Sx, Sy, Sz;  // shear of the watertight test
    int      kz;
    // plane distance along an axis: t = fma(float(q), r, cn) for the near plane (the box grown by the pad),
    // fma(float(q), r, cf) for the far one
    float    rx, ry, rz, cnx, cny, cnz, cfx, cfy, cfz;
    bool     negx, negy, negz;              // direction sign: the near plane is max instead of min
};

PRT_DEV float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

template <bool TOP = false>
PRT_DEV void ray_prepare(const BvhView& bvh, f3 o, float tmin, f3 d, float tmax, RayCtx& r)
{
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.tmin = tmin; r.tmax = tmax;
    // A ray with a NaN in its origin or direction (the reference keeps NaN normals for corners without one, Model.cpp:171-175)
    // hits nothing: every comparison of the triangle test fails. The slab test, however, drops NaNs (v_max / v_min return
    // the other operand), so such a ray would walk the whole tree first. An infinite tmin closes every box instead: same
    // result (a miss with t = tmax), no walk.
    if((o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (d.x != d.x) | (d.y != d.y) | (d.z != d.z)) r.tmin = __builtin_inff();
    int kz = 0;
    if(gabs(d.y) > gabs(sel3(d.x, d.y, d.z, kz))) kz = 1;
    if(gabs(d.z) > gabs(sel3(d.x, d.y, d.z, kz))) kz = 2;
    int kx = kz + 1; if(kx == 3) kx = 0;
    int ky = kx + 1; if(ky == 3) ky = 0;
    // Woop et al. swap kx/ky when d[kz] < 0 to keep the winding; with two-sided triangles the swap only
    // negates U, V, W, det and T together (every product is the same, every difference antisymmetric),
    // so t = T/det, u = V/det, v = W/det and the sign tests come out bit-identical without it.
    const float dkz = sel3(d.x, d.y, d.z, kz);
    r.kz = kz;
    if(!TOP)  // the top level of a two-level scene holds no triangles: the shear is set when an instance is entered
    {
        r.Sx = sel3(d.x, d.y, d.z, kx) / dkz;
        r.Sy = sel3(d.x, d.y, d.z, ky) / dkz;
        r.Sz = 1.0f / dkz;
    }
    float amax = bvh.sceneAbs;
    amax       = gmax(amax, gabs(o.x));
    amax       = gmax(amax, gabs(o.y));
    amax       = gmax(amax, gabs(o.z));
    float pad = 3.814697265625e-06f * amax;
    if(TOP) pad = pad * bvh.padScale;  // top level of a two-level scene
    // plane distance t = (qmin + q*qscale - (o +- pad)) * rd as one fma on float(q). What the pad has to cover, in units of
    // 2^-24 * amax along the axis (|plane - o| <= 2 amax): v_rcp_f32 (1 ulp, a common factor of both terms) 4, the
    // roundings of r, of the two sums and the product in c, and of the fma 2 + 1 + 2 + 2 + 2 = 13 in all, and the 16 by
    // which tri_core lets a hit point lie outside its triangle's extent: 29 of the pad's 64.
#define PRT_SLAB_SETUP(c, k, rr, cn, cf, ng)                         \
    {                                                                \
        float dk = d.c;                                              \
        if(gabs(dk) < 1e-20f) dk = (dk < 0.0f) ? -1e-20f : 1e-20f;   \
        const float rd  = __builtin_amdgcn_rcpf(dk);                 \
        ng              = dk < 0.0f;                                 \
        const float sp  = ng ? -pad : pad;                           \
        cn              = (bvh.qmin[k] - (o.c + sp)) * rd;           \
        cf              = (bvh.qmin[k] - (o.c - sp)) * rd;           \
        rr              = rd * bvh.qscale[k];                        \
    }
    PRT_SLAB_SETUP(x, 0, r.rx, r.cnx, r.cfx, r.negx)
    PRT_SLAB_SETUP(y, 1, r.ry, r.cny, r.cfy, r.negy)
    PRT_SLAB_SETUP(z, 2, r.rz, r.cnz, r.cfz, r.negz)
#undef PRT_SLAB_SETUP
}

// A vertex relative to the ray origin: the coordinate along the ray's major axis (kz), the two minor ones, and the minor
// ones after the shear of the watertight test.
struct TriVtx
{
    float kx, ky, kz;  // (q - o)[kx], [ky], [kz]
    float x, y;        // sheared: kx - Sx*kz, ky - Sy*kz
};
PRT_DEV TriVtx tri_vertex(const RayCtx& r, int kx, int ky, float4 q)
{
    const float dx = q.x - r.ox, dy = q.y - r.oy, dz = q.z - r.oz;
    TriVtx      v;
    v.kz = sel3(dx, dy, dz, r.kz);
    v.kx = sel3(dx, dy, dz, kx);
    v.ky = sel3(dx, dy, dz, ky);
    v.x  = v.kx - r.Sx * v.kz;
    v.y  = v.ky - r.Sy * v.kz;
    return v;
}
// The test proper, from the prepared corners and the three edge functions as fp32 evaluates them:
// U = C.x*B.y - C.y*B.x, V = A.x*C.y - A.y*C.x, W = B.x*A.y - B.y*A.x.
PRT_DEV bool tri_core(const RayCtx& r, float sceneAbs, const TriVtx& A, const TriVtx& B, const TriVtx& C, float U, float V,
                      float W, float& tOut, float& uOut, float& vOut)
{
    if(U == 0.0f || V == 0.0f || W == 0.0f)
    {
        const double CxBy = (double)C.x * (double)B.y, CyBx = (double)C.y * (double)B.x;
        U                 = (float)(CxBy - CyBx);
        const double AxCy = (double)A.x * (double)C.y, AyCx = (double)A.y * (double)C.x;
        V                 = (float)(AxCy - AyCx);
        const double BxAy = (double)B.x * (double)A.y, ByAx = (double)B.y * (double)A.x;
        W                 = (float)(BxAy - ByAx);
    }
    if((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = (U + V) + W;
    if(det == 0.0f)
        return false;
    const float Az  = r.Sz * A.kz;
    const float Bz  = r.Sz * B.kz;
    const float Cz  = r.Sz * C.kz;
    const float T   = (U * Az + V * Bz) + W * Cz;
    const float rcp = 1.0f / det;
    const float t   = T * rcp;
    if(!(t > r.tmin && t < r.tmax))
        return false;
    // A ray that grazes the triangle's plane (|d.n| < ~1e-3 |d|) can pass the edge tests with weights U, V, W that are
    // rounding noise: the hit point o + t*d then lies outside the triangle, and no conservative box test can be made
    // to keep such a hit. So it is not one: along the two minor axes the hit point, rebuilt from the same weights
    // (depth along the major axis = the weighted vertex depths, a convex combination and hence always inside the
    // triangle's extent; minor axes = shear * depth), has to lie within the triangle's extent grown by a quarter of
    // the pad of the box test. Found by tools/fuzz_parity.py (one pixel in ~10^8).
    {
        float amax = sceneAbs;
        amax       = gmax(amax, gabs(r.ox));
        amax       = gmax(amax, gabs(r.oy));
        amax       = gmax(amax, gabs(r.oz));
        const float hpad = 9.5367431640625e-07f * amax;  // 16 * 2^-24
        const float zh   = ((U * A.kz + V * B.kz) + W * C.kz) * rcp;
        const float xh = r.Sx * zh, yh = r.Sy * zh;
        if(xh < gmin(gmin(A.kx, B.kx), C.kx) - hpad || xh > gmax(gmax(A.kx, B.kx), C.kx) + hpad) return false;
        if(yh < gmin(gmin(A.ky, B.ky), C.ky) - hpad || yh > gmax(gmax(A.ky, B.ky), C.ky) + hpad) return false;
    }
    tOut = t;
    uOut = V * rcp;
    vOut = W * rcp;
    return true;
}
PRT_DEV bool tri_test(const RayCtx& r, float sceneAbs, float4 q0, float4 q1, float4 q2, float& tOut, float& uOut,
                      float& vOut)
{
    int kx = r.kz + 1; if(kx == 3) kx = 0;
    int ky = kx + 1; if(ky == 3) ky = 0;
    const TriVtx A = tri_vertex(r, kx, ky, q0), B = tri_vertex(r, kx, ky, q1), C = tri_vertex(r, kx, ky, q2);
    const float  U = C.x * B.y - C.y * B.x;
    const float  V = A.x * C.y - A.y * C.x;
    const float  W = B.x * A.y - B.y * A.x;
    return tri_core(r, sceneAbs, A, B, C, U, V, W, tOut, uOut, vOut);
}

// ---- per-lane stack: LDS [entry][thread] + scratch spill --------------------------------------------
#define PRT_STACK_AT(lds, i) (lds)[((i) + PRT_COLD_WORDS) * PRT_TRACE_BLOCK]  // the lane's cold words come first

template <int LS>
PRT_DEV int32_t stack_pop(int& sp, const uint32_t* lds, const int32_t* spill)
{
    --sp;
    int32_t v;
    if(__builtin_expect(sp < LS, 1))
        v = (int32_t)PRT_STACK_AT(lds, sp);
    else
    {
        v = spill[sp - LS];
        asm volatile("" : "+v"(v));  // keeps the two loads apart: ds_read / scratch_load, not one flat load
    }
    return v;
}
template <int LS>
PRT_DEV void stack_push_slow(int& sp, uint32_t* lds, int32_t* spill, int32_t v)
{
    if(sp < LS)
        PRT_STACK_AT(lds, sp) = (uint32_t)v;
    else
    {
        asm volatile("" ::: "memory");
        spill[sp - LS] = v;
    }
    ++sp;
}

#define PRT_CE(ta, ra, tb, rb)                     \
    {                                              \
        const bool    s_  = tb < ta;               \
        const float   tl_ = s_ ? tb : ta;          \
        const float   th_ = s_ ? ta : tb;          \
        const int32_t rl_ = s_ ? rb : ra;          \
        const int32_t rh_ = s_ ? ra : rb;          \
        ta = tl_; tb = th_; ra = rl_; rb = rh_;    \
    }

// One BVH4 node: four slab tests, nearest hit child becomes `cur`, the others are pushed (farthest
// first). No hit: pop. ANY rays skip the distance sort.
template <bool ANY, int LS>
PRT_DEV void node4_step(const BvhView& bvh, const RayCtx& r, float tbest, int32_t& cur, int& sp, uint32_t* lds,
                        int32_t* spill, const uint4* top)
{
    uint4 bx, by, bz, cq;
#if PRT_NODE_ASM
    {
        // Lanes below nTop read their node from the LDS copy, the others from memory, into the SAME registers. Written
        // by the compiler as if / else, the second group's loads wait for the first group's (a write-after-write on the
        // destination registers as far as its s_waitcnt insertion can tell), so a step pays the memory round trip AND
        // the LDS round trip. The lanes are disjoint, so there is no hazard: both groups' loads are issued back to back
        // under their own exec masks and waited for together.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4              a, b, c, d;
        unsigned long long sv;
        const uint32_t     off  = (uint32_t)cur << 6;
        const uint32_t     loff = off + (uint32_t)(uintptr_t)top;
        asm volatile("v_cmp_gt_u32_e32 vcc, %[ntop], %[cur]\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_andn2_b64 exec, exec, vcc\n\t"
                     "global_load_dwordx4 %[a], %[off], %[base]\n\t"
                     "global_load_dwordx4 %[b], %[off], %[base] offset:16\n\t"
                     "global_load_dwordx4 %[c], %[off], %[base] offset:32\n\t"
                     "global_load_dwordx4 %[d], %[off], %[base] offset:48\n\t"
                     "s_and_b64 exec, %[sv], vcc\n\t"
                     "ds_read_b128 %[a], %[loff]\n\t"
                     "ds_read_b128 %[b], %[loff] offset:16\n\t"
                     "ds_read_b128 %[c], %[loff] offset:32\n\t"
                     "ds_read_b128 %[d], %[loff] offset:48\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [sv] "=&s"(sv)
                     : [cur] "v"(cur), [ntop] "s"(bvh.nTop), [off] "v"(off), [base] "s"(bvh.nodes), [loff] "v"(loff)
                     : "vcc", "memory");
        bx = make_uint4(a.x, a.y, a.z, a.w);
        by = make_uint4(b.x, b.y, b.z, b.w);
        bz = make_uint4(c.x, c.y, c.z, c.w);
        cq = make_uint4(d.x, d.y, d.z, d.w);
    }
#else
    if(PRT_TOP_NODES && (uint32_t)cur < bvh.nTop)
    {
        // an LDS-typed pointer: ds_read_b128, not a flat load that the compiler would merge with the global one below
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const u32x4* lds_u4;
        lds_u4      tp = (lds_u4)(uintptr_t)(top + ((uint32_t)cur << 2));
        const u32x4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        bx = make_uint4(a.x, a.y, a.z, a.w);
        by = make_uint4(b.x, b.y, b.z, b.w);
        bz = make_uint4(c.x, c.y, c.z, c.w);
        cq = make_uint4(d.x, d.y, d.z, d.w);
    }
    else
    {
        const uint4* np = (const uint4*)((const char*)bvh.nodes + ((uint32_t)cur << 6));
        bx = np[0]; by = np[1]; bz = np[2]; cq = np[3];
    }
#endif
    const float  inf = __builtin_inff();
    // words: .x = min of children 0,1  .y = min of 2,3  .z = max of 0,1  .w = max of 2,3 (16 bits each)
    const uint32_t nx01 = r.negx ? bx.z : bx.x, nx23 = r.negx ? bx.w : bx.y;
    const uint32_t fx01 = r.negx ? bx.x : bx.z, fx23 = r.negx ? bx.y : bx.w;
    const uint32_t ny01 = r.negy ? by.z : by.x, ny23 = r.negy ? by.w : by.y;
    const uint32_t fy01 = r.negy ? by.x : by.z, fy23 = r.negy ? by.y : by.w;
    const uint32_t nz01 = r.negz ? bz.z : bz.x, nz23 = r.negz ? bz.w : bz.y;
    const uint32_t fz01 = r.negz ? bz.x : bz.z, fz23 = r.negz ? bz.y : bz.w;
#define PRT_QLO(w) ((float)((w) & 0xffffu))
#define PRT_QHI(w) ((float)((w) >> 16))
#define PRT_SLAB(Q, nxw, nyw, nzw, fxw, fyw, fzw, key)                                                               \
    {                                                                                                                \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(Q(nxw), r.rx, r.cnx),                       \
                                                         __builtin_fmaf(Q(nyw), r.ry, r.cny)),                      \
                                         __builtin_fmaxf(__builtin_fmaf(Q(nzw), r.rz, r.cnz), r.tmin));             \
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(Q(fxw), r.rx, r.cfx),                       \
                                                         __builtin_fmaf(Q(fyw), r.ry, r.cfy)),                      \
                                         __builtin_fminf(__builtin_fmaf(Q(fzw), r.rz, r.cfz), tbest));              \
        key = (tn <= tf) ? tn : inf;                                                                                 \
    }
    float t0, t1, t2, t3;
    PRT_SLAB(PRT_QLO, nx01, ny01, nz01, fx01, fy01, fz01, t0)
    PRT_SLAB(PRT_QHI, nx01, ny01, nz01, fx01, fy01, fz01, t1)
    PRT_SLAB(PRT_QLO, nx23, ny23, nz23, fx23, fy23, fz23, t2)
    PRT_SLAB(PRT_QHI, nx23, ny23, nz23, fx23, fy23, fz23, t3)
#undef PRT_SLAB
#undef PRT_QLO
#undef PRT_QHI
    int32_t r0 = (int32_t)cq.x, r1 = (int32_t)cq.y, r2 = (int32_t)cq.z, r3 = (int32_t)cq.w;
    if(!ANY)
    {
        PRT_CE(t0, r0, t1, r1)
        PRT_CE(t2, r2, t3, r3)
        PRT_CE(t0, r0, t2, r2)
        PRT_CE(t1, r1, t3, r3)
        PRT_CE(t1, r1, t2, r2)
    }
    const bool v0 = t0 < inf, v1 = t1 < inf, v2 = t2 < inf, v3 = t3 < inf;
    if(__builtin_expect(sp + 3 <= LS, 1))
    {
        PRT_STACK_AT(lds, sp) = (uint32_t)r3;
        sp += v3 ? 1 : 0;
        PRT_STACK_AT(lds, sp) = (uint32_t)r2;
        sp += v2 ? 1 : 0;
        PRT_STACK_AT(lds, sp) = (uint32_t)r1;
        sp += v1 ? 1 : 0;
    }
    else
    {
        if(v3) stack_push_slow<LS>(sp, lds, spill, r3);
        if(v2) stack_push_slow<LS>(sp, lds, spill, r2);
        if(v1) stack_push_slow<LS>(sp, lds, spill, r1);
    }
    if(v0)
        cur = r0;
    else
        cur = stack_pop<LS>(sp, lds, spill);
}

// All triangles of one leaf. Returns true when an ANY ray found an occluder.
// sceneAbs / primBase: of the structure the lane is in (single-level: bvh.sceneAbs, 0)
// Leaf reference = ~((slot << 3) | code); a slot is 64 bytes (one sector: what a divergent gather costs is the number of
// sectors it touches), a leaf over `count` triangles owns `count` consecutive slots. code = count: the triangles follow each
// other, 3 x float4 each (prim id in v0.w). code 3 (PRT_LEAF_QUAD, leaves of <= 2): two triangles that form a fan
// (A, B, C), (A, C, D) — the two halves of a quad, 9 leaves in 10 of a tessellated scene — stored as the four corners in ONE
// slot: rows (A | id0), (B | id1), (C | -), (D | -). Same floats into the same arithmetic as two separate tests: the corners
// are sheared once each, and the second triangle's third edge function B'.x*A'.y - B'.y*A'.x = C.x*A.y - C.y*A.x is
// exactly -(A.x*C.y - A.y*C.x), the first one's second (products commute, a - b = -(b - a) in IEEE arithmetic).
#define PRT_COLD_AT(lds, j) (lds)[(j) * PRT_TRACE_BLOCK]  // 0: tag, 1: prim, 2: u, 3: v
// COLD: prim / u / v of the best hit live in the lane's LDS words (trace_stream), only t stays in a register.
template <bool ANY, bool COLD>
PRT_DEV bool leaf_hit(Hit& best, uint32_t* lds, float tt, float uu, float vv, uint32_t p)
{
    if(ANY)
    {
        if(COLD)
            PRT_COLD_AT(lds, 1) = p;
        else
            best.prim = p;
        best.t = tt;
        return true;
    }
    bool closer = tt < best.t;
    if(tt == best.t) closer = p < (COLD ? PRT_COLD_AT(lds, 1) : best.prim);
    if(closer)
    {
        best.t = tt;
        if(COLD)
        {
            PRT_COLD_AT(lds, 1) = p;
            PRT_COLD_AT(lds, 2) = __float_as_uint(uu);
            PRT_COLD_AT(lds, 3) = __float_as_uint(vv);
        }
        else
        {
            best.u    = uu;
            best.v    = vv;
            best.prim = p;
        }
    }
    return false;
}
template <bool ANY, bool INSTR, bool COLD = false>
PRT_DEV bool leaf_test(const BvhView& bvh, const RayCtx& r, int32_t& leafRef, Hit& best, TraceCount& tc, float sceneAbs,
                       uint32_t primBase, uint32_t* lds = nullptr)
{
    const uint32_t ref   = ~(uint32_t)leafRef;
    leafRef              = kEmptyLeaf;  // what is left of the leaf after this call (PRT_LEAF_SPLIT_PAIR)
    const uint32_t first = ref >> 3, count = ref & 7u;
    if(count == 0) return false;
    const float4* t = bvh.tris + (size_t)first * PRT_LEAF_SLOT_F4;  // the leaf's first 64-byte slot (k_emit_leaves)
#if PRT_LEAF_QUAD
    {
        // count is 1, 2 or 3 (fan). One code path for all three, so that a wave whose lanes hold different kinds of leaf
        // runs two tests, not three: the first triangle is rows 0-2 in every form; the second one's corners are prepared
        // per form (the short divergent part), its test is shared.
        const float4 qa = t[0], qb = t[1], qc = t[2], qd = t[3];
        int kx = r.kz + 1; if(kx == 3) kx = 0;
        int ky = kx + 1; if(ky == 3) ky = 0;
        const TriVtx A = tri_vertex(r, kx, ky, qa), B = tri_vertex(r, kx, ky, qb), C = tri_vertex(r, kx, ky, qc);
        const float  U0 = C.x * B.y - C.y * B.x;
        const float  V0 = A.x * C.y - A.y * C.x;
        const float  W0 = B.x * A.y - B.y * A.x;
        float        tt, uu, vv;
        if(INSTR) tc.tris++;
        if(tri_core(r, sceneAbs, A, B, C, U0, V0, W0, tt, uu, vv))
            if(leaf_hit<ANY, COLD>(best, lds, tt, uu, vv, primBase + (uint32_t)__float_as_int(qa.w))) return true;
        if(count == 1u) return false;
#if PRT_LEAF_SPLIT_PAIR == 1
        // every ray hands the second triangle of a plain pair back
        if(count == 2u)
        {
            leafRef = (int32_t) ~(((first + 1u) << 3) | 1u);
            return false;
        }
        const TriVtx   A1 = A, B1 = C, C1 = tri_vertex(r, kx, ky, qd);
        const float    W1  = -V0;
        const uint32_t id1 = (uint32_t)__float_as_int(qb.w);
#else
        // 2: any-hit rays only — a closest-hit ray wants the second triangle now (a nearer hit culls what it visits next),
        // an any-hit ray only wants an occluder; 3: every ray, with the code of the unsplit form behind it (measurement)
        if((PRT_LEAF_SPLIT_PAIR == 3 || (PRT_LEAF_SPLIT_PAIR == 2 && ANY)) && count == 2u)
        {
            leafRef = (int32_t) ~(((first + 1u) << 3) | 1u);
            return false;
        }
        TriVtx   A1, B1, C1;
        float    W1;
        uint32_t id1;
        if(count == 3u)
        {
            A1  = A;
            B1  = C;
            C1  = tri_vertex(r, kx, ky, qd);
            W1  = -V0;
            id1 = (uint32_t)__float_as_int(qb.w);
        }
        else
        {
            // the second triangle of a plain pair: packed behind the first one, or in a slot of its own (k_emit_leaves)
            const float4* t2 = t + (PRT_LEAF_SPLIT_PAIR ? PRT_LEAF_SLOT_F4 : 3);
            const float4  qe = t2[0], qf = t2[1], qg = t2[2];
            A1  = tri_vertex(r, kx, ky, qe);
            B1  = tri_vertex(r, kx, ky, qf);
            C1  = tri_vertex(r, kx, ky, qg);
            W1  = B1.x * A1.y - B1.y * A1.x;
            id1 = (uint32_t)__float_as_int(qe.w);
        }
#endif
        const float U1 = C1.x * B1.y - C1.y * B1.x;
        const float V1 = A1.x * C1.y - A1.y * C1.x;
        if(INSTR) tc.tris++;
        if(tri_core(r, sceneAbs, A1, B1, C1, U1, V1, W1, tt, uu, vv))
            if(leaf_hit<ANY, COLD>(best, lds, tt, uu, vv, primBase + id1)) return true;
        return false;
    }
#else
#if PRT_TRI_PREFETCH
    float4        a = t[0], b = t[1], c = t[2];
#endif
    for(uint32_t i = 0; i < count; ++i)
    {
#if PRT_TRI_PREFETCH
        // the next triangle's loads are in flight while this one is tested
        float4 na = a, nb = b, nc = c;
        if(i + 1 < count)
        {
            na = t[3 * i + 3];
            nb = t[3 * i + 4];
            nc = t[3 * i + 5];
        }
#else
        const float4 a = t[3 * i], b = t[3 * i + 1], c = t[3 * i + 2];
#endif
        float tt, uu, vv;
        if(INSTR) tc.tris++;
        if(tri_test(r, sceneAbs, a, b, c, tt, uu, vv))
            if(leaf_hit<ANY, COLD>(best, lds, tt, uu, vv, primBase + (uint32_t)__float_as_int(a.w))) return true;
#if PRT_TRI_PREFETCH
        a = na;
        b = nb;
        c = nc;
#endif
    }
    return false;
#endif
}

PRT_DEV Hit hit_none(float tmax)
{
    Hit h;
    h.prim = 0xffffffffu;
    h.u = h.v = 0.0f;
    h.t       = tmax;
    return h;
}

// Copies the top of the tree into the block's LDS (behind the stacks) and returns it; all threads of the block call it.
template <int LS = PRT_LDS_STACK>
PRT_DEV const uint4* stage_top_nodes(const BvhView& bvh, uint32_t* ldsBase)
{
    uint4* top = (uint4*)(ldsBase + (LS + PRT_COLD_WORDS) * PRT_TRACE_BLOCK);
    if(PRT_TOP_NODES)
    {
        for(uint32_t k = threadIdx.x; k < bvh.nTop * 4u; k += PRT_TRACE_BLOCK) top[k] = bvh.nodes[k];
        __syncthreads();
    }
    return top;
}

// ---- two-level scenes: entering and leaving an instance ------------------------------------------------
// The lane's ray in the object space of the instance behind top-level leaf `ref`, its slab constants on that
// structure's grid; returns the structure's root reference (an empty leaf for an empty slot).
PRT_DEV int32_t instance_enter(const BvhView& bvh, int32_t ref, f3 o, float tmin, f3 d, float tmax, RayCtx& r,
                               float& curAbs, uint32_t& primBase)
{
    const uint32_t lref = ~(uint32_t)ref;
    if((lref & 7u) == 0u) return kEmptyLeaf;
    const float4* rec = bvh.inst + (size_t)(lref >> 3) * PRT_INST_F4;
    const float4  w0 = rec[0], w1 = rec[1], w2 = rec[2], g0 = rec[3], g1 = rec[4], mt = rec[5];
    if(!((uint32_t)__float_as_int(mt.y) & 1u))
    {
        // o' = (W0*o.x + W1*o.y) + (W2*o.z + W3), d' = (W0*d.x + W1*d.y) + W2*d.z: the oracle's rayToObject
        const f3 oo = mk3((w0.x * o.x + w0.y * o.y) + (w0.z * o.z + w0.w), (w1.x * o.x + w1.y * o.y) + (w1.z * o.z + w1.w),
                          (w2.x * o.x + w2.y * o.y) + (w2.z * o.z + w2.w));
        const f3 od = mk3((w0.x * d.x + w0.y * d.y) + w0.z * d.z, (w1.x * d.x + w1.y * d.y) + w1.z * d.z,
                          (w2.x * d.x + w2.y * d.y) + w2.z * d.z);
        o = oo;
        d = od;
    }
    BvhView ob   = bvh;
    ob.qmin[0]   = g0.x; ob.qmin[1] = g0.y; ob.qmin[2] = g0.z;
    ob.qscale[0] = g1.x; ob.qscale[1] = g1.y; ob.qscale[2] = g1.z;
    ob.sceneAbs  = g0.w;
    ray_prepare(ob, o, tmin, d, tmax, r);
    curAbs   = g0.w;
    primBase = (uint32_t)__float_as_int(mt.x);
    return __float_as_int(g1.w);
}

// ---- one ray per lane, all lanes start together (AO rays inside k_ao_shade) ---------------------------
// ANY = true: terminate-on-first-hit / skip-closest-hit-shader occlusion query (prim != ~0u on hit).
// INSTR = true additionally counts node visits and triangle tests (profiling level 2).
template <bool ANY, bool INSTR = false, bool TWO = false>
PRT_DEV Hit trace_ray(const BvhView& bvh, uint32_t* lds, const uint4* top, f3 o, float tmin, f3 d, float tmax,
                      TraceCount* tcp = nullptr)
{
    constexpr int LS = PRT_LS(TWO);
    RayCtx r;
    ray_prepare<TWO>(bvh, o, tmin, d, tmax, r);
    Hit        best = hit_none(tmax);
    TraceCount tc   = {};
    int32_t    spill[TWO ? PRT_SPILL_STACK_TWO : PRT_SPILL_STACK];
    PRT_STACK_AT(lds, 0) = (uint32_t)kSentinel;
    int      sp       = 1;
    int32_t  cur      = bvh.root;
    float    curAbs   = bvh.sceneAbs;
    uint32_t primBase = 0;
    bool     inInst   = false;
    while(cur != kSentinel)
    {
        if(TWO)
        {
            if(cur == kInstExit)
            {
                ray_prepare<true>(bvh, o, tmin, d, tmax, r);
                inInst = false;
                cur    = stack_pop<LS>(sp, lds, spill);
                continue;
            }
            if(cur < 0 && !inInst)
            {
                const int32_t root = instance_enter(bvh, cur, o, tmin, d, tmax, r, curAbs, primBase);
                if(root == kEmptyLeaf)
                    cur = stack_pop<LS>(sp, lds, spill);
                else
                {
                    stack_push_slow<LS>(sp, lds, spill, kInstExit);
                    inInst = true;
                    cur    = root;
                }
                continue;
            }
        }
        if(cur >= 0)
        {
            if(INSTR) tc.nodes++;
            node4_step<ANY, LS>(bvh, r, best.t, cur, sp, lds, spill, top);
        }
        else
        {
            if(leaf_test<ANY, INSTR>(bvh, r, cur, best, tc, TWO ? curAbs : bvh.sceneAbs, TWO ? primBase : 0u)) break;
            if(cur == kEmptyLeaf) cur = stack_pop<LS>(sp, lds, spill);  // else: the rest of the leaf
        }
    }
    if(INSTR && tcp)
    {
        tcp->nodes += tc.nodes;
        tcp->tris += tc.tris;
    }
    return best;
}

#ifdef PRT_TIMELINE
// measurement builds (tools/timeline.py): per wave of the grid, wall-clock ticks (100 MHz) at the start of its last stream, when it
// found the queue empty, when it left, and the rays it traced
static __device__ unsigned long long g_timeline[8192 * 4];
#endif
// ---- persistent wave: a stream of `count` rays ----------------------------------------------------------
// IO::load(i, o, tmin, d, tmax, tag) -> bool (false: a placeholder that does not count in the statistics);
// IO::store(tag, hit) is called once per ray when it has finished.
// TWO: two-level scene. A lane at the top level that reaches a leaf (= an instance) enters it at the next node step;
// a lane that pops the exit marker leaves the instance once its postponed leaf (which belongs to the instance) has
// been tested. The world-space ray stays in seven registers (the TWO kernels are compiled for 6 waves per SIMD, 80 VGPRs, and have them).
template <bool ANY, bool INSTR, class IO, bool TWO = false>
PRT_DEV void trace_stream(const BvhView& bvh, uint32_t* lds, const uint4* top, uint32_t count, uint32_t* workCtr,
                          const TraceTune tune, IO& io, uint32_t& nskipped, TraceCount& tc)
{
    constexpr int            LS     = PRT_LS(TWO);
    const int                lane   = threadIdx.x & 63;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    uint32_t                 chunkPos = 0, chunkEnd = 0;  // wave-uniform
    bool                     exhausted = false;
    bool                     active    = false;
    int32_t                  cur = kSentinel, leaf = kEmptyLeaf;
    int                      sp = 1;
    int32_t                  spill[TWO ? PRT_SPILL_STACK_TWO : PRT_SPILL_STACK];
    RayCtx                   r;
    Hit                      best = hit_none(0.0f);  // only best.t is used: prim / u / v are PRT_COLD_AT words
    uint32_t                 nskip = 0;              // wave-uniform
    // two-level scenes: the structure the lane is in
    float                    curAbs   = bvh.sceneAbs;
    uint32_t                 primBase = 0;
    bool                     inInst   = false;
    f3                       wo = mk3(0.0f), wd = mk3(0.0f);  // the ray in world space
    float                    wtmin = 0.0f;                    // (r.tmin may have been closed by a NaN in object space)
    constexpr int32_t        kStepEnd = TWO ? kInstExit : kSentinel;  // references below this are inner nodes
    r.tmin = r.tmax = 0.0f;
    PRT_STACK_AT(lds, 0) = (uint32_t)kSentinel;  // bottom of the lane's stack; sp >= 1 always

#ifdef PRT_TIMELINE
    const unsigned long long tl0 = wall_clock64();
    unsigned long long       tlE = 0;
    uint32_t                 tlRays = 0;
#endif
    unsigned long long tStart = 0, tMark = 0;
    if(INSTR) tStart = tMark = __builtin_amdgcn_s_memtime();
#define PRT_LAP(field)                                                  \
    if(INSTR)                                                           \
    {                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        tc.field += (uint32_t)(now_ - tMark);                           \
        tMark = now_;                                                   \
    }
    // Every trip of this loop starts rays, advances them or finishes them, and a wave whose queue is exhausted and whose
    // lanes are idle leaves through the `break` below. The bound is a safety net against a corrupted structure (a cycle
    // in the node links): far above what any ray set needs (2^26 trips of >= 1 step); a wave that reaches it says so
    // in tc.overrun, which the kernels add to the context's statistics (prt_stats.trace_overruns).
    bool overrun = true;
    for(uint32_t guard = 0; guard < (1u << 26); ++guard)
    {
        // ---- refill idle lanes from the queue
        const unsigned long long idleMask = __ballot(!active);
        const uint32_t           nIdle    = (uint32_t)__popcll(idleMask);
        if(!exhausted && nIdle >= tune.refillMin)
        {
            if(chunkPos == chunkEnd)
            {
                uint32_t b = 0;
                if(lane == 0) b = atomicAdd(workCtr, PRT_CHUNK);
                b        = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                chunkPos = b < count ? b : count;
                chunkEnd = (count - chunkPos) < PRT_CHUNK ? count : chunkPos + PRT_CHUNK;
                if(chunkPos == chunkEnd) exhausted = true;
#ifdef PRT_TIMELINE
                if(exhausted) tlE = wall_clock64();
#endif
            }
            const uint32_t avail = chunkEnd - chunkPos;
            const uint32_t take  = nIdle < avail ? nIdle : avail;
            const uint32_t rank  = (uint32_t)__popcll(idleMask & ltMask);
            if(INSTR && lane == 0 && take)
            {
                tc.wSetup++;
                tc.lSetup += take;
            }
            bool counted = true;
            if(!active && rank < take)
            {
                f3       o, d;
                float    tmin, tmax;
                uint32_t tag;
                counted = io.load(chunkPos + rank, o, tmin, d, tmax, tag);
                ray_prepare<TWO>(bvh, o, tmin, d, tmax, r);
                best.t = tmax;
                PRT_COLD_AT(lds, 0) = tag;
                PRT_COLD_AT(lds, 1) = 0xffffffffu;
                PRT_COLD_AT(lds, 2) = 0u;
                PRT_COLD_AT(lds, 3) = 0u;
                cur    = bvh.root;
                leaf   = kEmptyLeaf;
                sp     = 1;
                active = true;
                if(TWO)
                {
                    inInst = false;
                    wo     = o;
                    wd     = d;
                    wtmin  = tmin;
                }
            }
            nskip += (uint32_t)__popcll(__ballot(!counted));
            chunkPos += take;
#ifdef PRT_TIMELINE
            tlRays += take;
#endif
        }
        const uint32_t nActive = (uint32_t)__popcll(__ballot(active));
        PRT_LAP(cSetup)
        if(nActive == 0)
        {
            if(exhausted)
            {
                overrun = false;
                break;
            }
            continue;
        }

        // ---- node phase: every lane at an inner node steps; a lane that reaches its first leaf postpones
        //      it and keeps descending; lanes at a second leaf (or finished) wait
        for(;;)
        {
            if(INSTR)
            {
                const uint32_t n = (uint32_t)__popcll(__ballot(active && (uint32_t)cur < (uint32_t)kStepEnd));
                if(lane == 0 && n)
                {
                    tc.wNode++;
                    tc.lNode += n;
                }
            }
            if(active && (uint32_t)cur < (uint32_t)kStepEnd)
            {
                if(INSTR) tc.nodes++;
                node4_step<ANY, LS>(bvh, r, best.t, cur, sp, lds, spill, top);
                if(cur < 0 && leaf == kEmptyLeaf && (!TWO || inInst))
                {
                    leaf = cur;
                    cur  = stack_pop<LS>(sp, lds, spill);
                }
            }
            // lanes that can go on in this phase: at an inner node (two-level: lanes about to enter or leave an instance
            // wait for the transition phase below, like lanes at a second leaf wait for the triangle phase)
            const uint32_t nProd = (uint32_t)__popcll(__ballot(active && (uint32_t)cur < (uint32_t)kStepEnd));
            if(nProd == 0 || nActive - nProd > tune.stallMax) break;
        }

        PRT_LAP(cNode)
        // ---- triangle phase
        if(active && leaf == kEmptyLeaf && cur < 0 && (!TWO || inInst))
        {
            leaf = cur;
            cur  = stack_pop<LS>(sp, lds, spill);
        }
        for(uint32_t round = 0;; ++round)
        {
            const uint32_t nLeaf = (uint32_t)__popcll(__ballot(active && leaf != kEmptyLeaf));
            if(nLeaf == 0 || (round > 0 && nLeaf < tune.leafMin)) break;
            const bool has = active && leaf != kEmptyLeaf;
            if(INSTR)
            {
                // the wave runs as many sequential tests as its largest leaf; lTri counts the tests that did work
                uint32_t cnt = has ? (~(uint32_t)leaf & 7u) : 0u;
                if(PRT_LEAF_QUAD && cnt == 3u) cnt = 2u;
                if((PRT_LEAF_SPLIT_PAIR == 1 || PRT_LEAF_SPLIT_PAIR == 3 || (PRT_LEAF_SPLIT_PAIR == 2 && ANY)) && cnt == 2u && (~(uint32_t)leaf & 7u) == 2u) cnt = 1u;
                uint32_t       tot = cnt, mx = cnt;
                for(int o = 32; o > 0; o >>= 1)
                {
                    tot += (uint32_t)__shfl_xor((int)tot, o, 64);
                    const uint32_t m = (uint32_t)__shfl_xor((int)mx, o, 64);
                    mx               = m > mx ? m : mx;
                }
                if(lane == 0 && tot)
                {
                    tc.wTri += mx;
                    tc.lTri += tot;
                }
            }
            if(has)
            {
                const bool occluded = leaf_test<ANY, INSTR, true>(bvh, r, leaf, best, tc, TWO ? curAbs : bvh.sceneAbs, TWO ? primBase : 0u, lds);
                if(ANY && occluded)
                {
                    leaf = kEmptyLeaf;
                    cur  = kSentinel;
                    sp   = 1;
                }
                else if(leaf == kEmptyLeaf && cur < 0)
                {
                    leaf = cur;
                    cur  = stack_pop<LS>(sp, lds, spill);
                }
            }
        }

        // ---- two-level scenes: instance boundaries, for all lanes that wait at one (leave, then possibly enter the next)
        if(TWO && active)
        {
            // r.tmax is the ray's own in every space
            if(cur == kInstExit && leaf == kEmptyLeaf)
            {
                ray_prepare<true>(bvh, wo, wtmin, wd, r.tmax, r);
                inInst = false;
                cur    = stack_pop<LS>(sp, lds, spill);
            }
            if(cur < 0 && !inInst)
            {
                const int32_t root = instance_enter(bvh, cur, wo, wtmin, wd, r.tmax, r, curAbs, primBase);
                if(root == kEmptyLeaf)
                    cur = stack_pop<LS>(sp, lds, spill);
                else
                {
                    stack_push_slow<LS>(sp, lds, spill, kInstExit);
                    inInst = true;
                    cur    = root;
                }
            }
        }

        // ---- finished rays
        if(active && cur == kSentinel && leaf == kEmptyLeaf)
        {
            Hit h;
            h.prim = PRT_COLD_AT(lds, 1);
            h.u    = __uint_as_float(PRT_COLD_AT(lds, 2));
            h.v    = __uint_as_float(PRT_COLD_AT(lds, 3));
            h.t    = best.t;
            io.store(PRT_COLD_AT(lds, 0), h);
            active = false;
        }
        PRT_LAP(cTri)
    }
#ifdef PRT_TIMELINE
    if(lane == 0)
    {
        const uint32_t w = (blockIdx.x * (PRT_TRACE_BLOCK / 64u) + (threadIdx.x >> 6)) & 8191u;
        g_timeline[w * 4 + 0] = tl0;
        g_timeline[w * 4 + 1] = tlE;
        g_timeline[w * 4 + 2] = wall_clock64();
        g_timeline[w * 4 + 3] = tlRays;
    }
#endif
    if(lane == 0) nskipped += nskip;  // the callers sum it over the wave
    if(overrun) tc.overrun = 1u;
    if(INSTR) tc.cTotal += (uint32_t)(__builtin_amdgcn_s_memtime() - tStart);
#undef PRT_LAP
}

}  // namespace prt
