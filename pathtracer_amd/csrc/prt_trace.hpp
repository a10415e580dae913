// prt_trace.hpp — wave64 BVH traversal for gfx950: replaces traceNV / the RT cores
// (shaders/pathRT.rgen:692,819-822,963; shaders/AO.rgen:215,259-262).
//
// * BVH2, 64-byte nodes that hold BOTH child boxes: one node visit = four 16-byte loads from one
//   64-byte line (half of a 128-byte HBM/L2 line, so sibling-order BFS layout pairs them).
// * Leaves reference runs of <= 4 pre-gathered triangles, 48 bytes each (3 x float4, prim id in
//   v0.w), laid out in leaf order so a leaf is one contiguous 48..192-byte read.
// * Per-lane traversal stack: the first PRT_LDS_STACK entries live in LDS, laid out
//   [entry][lane] so that a wave's push/pop is one conflict-free ds_write_b32/ds_read_b32;
//   deeper entries (rare: the SAH trees built here are 20-35 levels deep) spill to scratch.
// * Watertight ray/triangle test (Woop, Benthin, Wald 2013) with the fp64 fallback for edge
//   functions that evaluate to exactly 0; two-sided triangles (TRIANGLE_CULL_DISABLE,
//   vkRT_TLAS.cpp:218-223); hit iff tmin < t < tmax; closest hit ties -> lower primitive id.
// * Conservative slab test: boxes grown by 64*2^-24*max(|origin|, |scene|), distances widened by 4 ulp.
#pragma once
#include "prt_device.hpp"

namespace prt {

#define PRT_LDS_STACK 24
#define PRT_SPILL_STACK 40
#define PRT_MAX_DEPTH (PRT_LDS_STACK + PRT_SPILL_STACK)
#define PRT_TRACE_BLOCK 256

struct alignas(16) Node64
{
    float   lmin[3];
    int32_t left;  // >= 0: inner node index; < 0: ~((first << 3) | count)
    float   lmax[3];
    int32_t right;
    float   rmin[3];
    uint32_t pad0;
    float   rmax[3];
    uint32_t pad1;
};
static_assert(sizeof(Node64) == 64, "node is one 64-byte line");

struct BvhView
{
    const float4* nodes;  // Node64 as 4 x float4
    const float4* tris;   // 3 x float4 per triangle, leaf order
    int32_t       root;   // node index, or a leaf reference when the scene has <= 4 triangles
    float         sceneAbs;
};

struct Hit
{
    uint32_t prim;
    float    u, v, t;
};

struct RayPre
{
    float ox, oy, oz;
    float Sx, Sy, Sz;
    int   kx, ky, kz;
    float onx, ony, onz, ofx, ofy, ofz;  // padded origins for near / far planes
    float rnx, rny, rnz, rfx, rfy, rfz;  // widened reciprocal directions
    bool  negx, negy, negz;
};

PRT_DEV float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

PRT_DEV void ray_prepare(const BvhView& bvh, f3 o, f3 d, RayPre& r)
{
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    int kz = 0;
    if(gabs(d.y) > gabs(sel3(d.x, d.y, d.z, kz))) kz = 1;
    if(gabs(d.z) > gabs(sel3(d.x, d.y, d.z, kz))) kz = 2;
    int kx = kz + 1; if(kx == 3) kx = 0;
    int ky = kx + 1; if(ky == 3) ky = 0;
    const float dkz = sel3(d.x, d.y, d.z, kz);
    if(dkz < 0.0f) { const int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = sel3(d.x, d.y, d.z, kx) / dkz;
    r.Sy = sel3(d.x, d.y, d.z, ky) / dkz;
    r.Sz = 1.0f / dkz;
    float amax = bvh.sceneAbs;
    amax       = gmax(amax, gabs(o.x));
    amax       = gmax(amax, gabs(o.y));
    amax       = gmax(amax, gabs(o.z));
    const float pad = 3.814697265625e-06f * amax;
#define PRT_SLAB_SETUP(c, on, of, rn, rf, ng)                        \
    {                                                                \
        float dk = d.c;                                              \
        if(gabs(dk) < 1e-20f) dk = (dk < 0.0f) ? -1e-20f : 1e-20f;   \
        const float rd = 1.0f / dk;                                  \
        ng             = dk < 0.0f;                                  \
        const float sp = ng ? -pad : pad;                            \
        on             = o.c + sp;                                   \
        of             = o.c - sp;                                   \
        rn             = rd * 0.99999976f;                           \
        rf             = rd * 1.00000024f;                           \
    }
    PRT_SLAB_SETUP(x, r.onx, r.ofx, r.rnx, r.rfx, r.negx)
    PRT_SLAB_SETUP(y, r.ony, r.ofy, r.rny, r.rfy, r.negy)
    PRT_SLAB_SETUP(z, r.onz, r.ofz, r.rnz, r.rfz, r.negz)
#undef PRT_SLAB_SETUP
}

PRT_DEV bool slab_test(const RayPre& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tmin,
                       float tbest, float& tnOut)
{
    float tn = tmin, tf = tbest;
    {
        const float pn = r.negx ? mxx : mnx, pf = r.negx ? mnx : mxx;
        tn = gmax(tn, (pn - r.onx) * r.rnx);
        tf = gmin(tf, (pf - r.ofx) * r.rfx);
    }
    {
        const float pn = r.negy ? mxy : mny, pf = r.negy ? mny : mxy;
        tn = gmax(tn, (pn - r.ony) * r.rny);
        tf = gmin(tf, (pf - r.ofy) * r.rfy);
    }
    {
        const float pn = r.negz ? mxz : mnz, pf = r.negz ? mnz : mxz;
        tn = gmax(tn, (pn - r.onz) * r.rnz);
        tf = gmin(tf, (pf - r.ofz) * r.rfz);
    }
    tnOut = tn;
    return tn <= tf;
}

PRT_DEV bool tri_test(const RayPre& r, float4 q0, float4 q1, float4 q2, float tmin, float tmax, float& tOut,
                      float& uOut, float& vOut)
{
    const float Ax_ = q0.x - r.ox, Ay_ = q0.y - r.oy, Az_ = q0.z - r.oz;
    const float Bx_ = q1.x - r.ox, By_ = q1.y - r.oy, Bz_ = q1.z - r.oz;
    const float Cx_ = q2.x - r.ox, Cy_ = q2.y - r.oy, Cz_ = q2.z - r.oz;
    const float Akz = sel3(Ax_, Ay_, Az_, r.kz), Bkz = sel3(Bx_, By_, Bz_, r.kz), Ckz = sel3(Cx_, Cy_, Cz_, r.kz);
    const float Ax = sel3(Ax_, Ay_, Az_, r.kx) - r.Sx * Akz;
    const float Ay = sel3(Ax_, Ay_, Az_, r.ky) - r.Sy * Akz;
    const float Bx = sel3(Bx_, By_, Bz_, r.kx) - r.Sx * Bkz;
    const float By = sel3(Bx_, By_, Bz_, r.ky) - r.Sy * Bkz;
    const float Cx = sel3(Cx_, Cy_, Cz_, r.kx) - r.Sx * Ckz;
    const float Cy = sel3(Cx_, Cy_, Cz_, r.ky) - r.Sy * Ckz;
    float       U  = Cx * By - Cy * Bx;
    float       V  = Ax * Cy - Ay * Cx;
    float       W  = Bx * Ay - By * Ax;
    if(U == 0.0f || V == 0.0f || W == 0.0f)
    {
        const double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U                 = (float)(CxBy - CyBx);
        const double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V                 = (float)(AxCy - AyCx);
        const double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W                 = (float)(BxAy - ByAx);
    }
    if((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = (U + V) + W;
    if(det == 0.0f)
        return false;
    const float Az  = r.Sz * Akz;
    const float Bz  = r.Sz * Bkz;
    const float Cz  = r.Sz * Ckz;
    const float T   = (U * Az + V * Bz) + W * Cz;
    const float rcp = 1.0f / det;
    const float t   = T * rcp;
    if(!(t > tmin && t < tmax))
        return false;
    tOut = t;
    uOut = V * rcp;
    vOut = W * rcp;
    return true;
}

// Per-lane stack: LDS [entry][thread] + scratch spill.
struct LaneStack
{
    uint32_t* lds;  // &stackLds[threadIdx.x]
    int32_t   spill[PRT_SPILL_STACK];
    int       sp;
    PRT_DEV void init(uint32_t* base) { lds = base; sp = 0; }
    PRT_DEV void push(int32_t v)
    {
        if(sp < PRT_LDS_STACK)
            lds[sp * PRT_TRACE_BLOCK] = (uint32_t)v;
        else
            spill[sp - PRT_LDS_STACK] = v;
        ++sp;
    }
    PRT_DEV int32_t pop()
    {
        --sp;
        return (sp < PRT_LDS_STACK) ? (int32_t)lds[sp * PRT_TRACE_BLOCK] : spill[sp - PRT_LDS_STACK];
    }
};

// ANY = true: terminate-on-first-hit / skip-closest-hit-shader occlusion query (returns prim != ~0u on hit).
struct TraceCount
{
    uint32_t nodes, tris;
};

// INSTR = true additionally counts node visits and triangle tests (profiling level 2: the N_node / N_tri
// of the algorithmic-bytes formula in DESIGN.md).
template <bool ANY, bool INSTR = false>
PRT_DEV Hit trace_ray(const BvhView& bvh, uint32_t* ldsStack, f3 o, float tmin, f3 d, float tmax,
                      TraceCount* tc = nullptr)
{
    RayPre r;
    ray_prepare(bvh, o, d, r);
    Hit best;
    best.prim = 0xffffffffu;
    best.u = best.v = 0.0f;
    best.t          = tmax;
    LaneStack st;
    st.init(ldsStack);
    int32_t cur = bvh.root;
    for(;;)
    {
        if(cur >= 0)
        {
            const float4* n  = bvh.nodes + (size_t)cur * 4;
            if(INSTR) tc->nodes++;
            const float4  q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            float         tl, tr;
            const bool    hl = slab_test(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, tmin, best.t, tl);
            const bool    hr = slab_test(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, tmin, best.t, tr);
            const int32_t li = __float_as_int(q0.w), ri = __float_as_int(q1.w);
            if(hl && hr)
            {
                const bool rightFirst = tr < tl;
                st.push(rightFirst ? li : ri);
                cur = rightFirst ? ri : li;
                continue;
            }
            if(hl) { cur = li; continue; }
            if(hr) { cur = ri; continue; }
        }
        else
        {
            const uint32_t ref   = ~(uint32_t)cur;
            const uint32_t first = ref >> 3, count = ref & 7u;
            const float4*  t     = bvh.tris + (size_t)first * 3;
            for(uint32_t i = 0; i < count; ++i)
            {
                const float4 a = t[3 * i], b = t[3 * i + 1], c = t[3 * i + 2];
                float        tt, uu, vv;
                if(INSTR) tc->tris++;
                if(tri_test(r, a, b, c, tmin, tmax, tt, uu, vv))
                {
                    const uint32_t p = (uint32_t)__float_as_int(a.w);
                    if(ANY)
                    {
                        best.prim = p;
                        best.t    = tt;
                        return best;
                    }
                    if(tt < best.t || (tt == best.t && p < best.prim))
                    {
                        best.t    = tt;
                        best.u    = uu;
                        best.v    = vv;
                        best.prim = p;
                    }
                }
            }
        }
        if(st.sp == 0)
            break;
        cur = st.pop();
    }
    return best;
}

}  // namespace prt
