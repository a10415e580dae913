// prt_bvh.hpp — binned-SAH BVH build on the GPU: replaces the BLAS/TLAS construction that the
// reference hands to vkCmdBuildAccelerationStructureNV (src/vkRTX_setup.cpp:84-132,138-186,391-415;
// external/NVIDIA_RTX/vkRT_BLAS.cpp:271, vkRT_TLAS.cpp:262). Opaque, two-sided triangles. The reference's scene — one
// identity instance of model 0 — is a single tree over the scene's triangles; with prt_set_instances the same builder
// makes one tree per bottom-level range and one over the instances' world boxes (`given`, leaf size 1).
//
// Algorithm: breadth-first, one pass of kernels per tree level, 16 bins per axis, leaves <= PRT_LEAF_MAX (2)
// triangles. Everything that decides the tree is order-independent: bin counts are integer
// atomics, bin / node / centroid bounds are integer atomicMin/Max on order-preserving float keys,
// SAH costs are evaluated per node in a fixed order, child slots and the stable partition come
// from prefix sums. The same input therefore always yields the same tree, bit for bit.
#pragma once
#include "prt_trace.hpp"

namespace prt {

#ifndef PRT_BINS
#define PRT_BINS 16  // measured on the bench scene (tools/bvh_quality.py): 8 / 16 / 32 / 64 bins give 13.0 / 13.0 / 13.2 / 13.2 node visits per ray
#endif
constexpr int      kBins    = PRT_BINS;
#ifndef PRT_LEAF_MAX
#define PRT_LEAF_MAX 2  // measured on the bench scene: 4 -> 2 turns 4.4 triangle tests per ray into 2.5 for +0.5 node visits (the BVH4 collapse absorbs the extra binary levels): traversal kernels -7 %; 1 and 3 are no better, 6 is 9 % slower
#endif
constexpr uint32_t kLeafMax = PRT_LEAF_MAX;
constexpr uint32_t kBinWords = 3 * kBins * 7;  // per splitting node: cnt[3][16], min[3][16][3], max[3][16][3]

struct BuildNode  // bounds as ordered keys while building
{
    uint32_t bmin[3];
    uint32_t first;
    uint32_t bmax[3];
    uint32_t count;
};
struct SplitRec
{
    int32_t  axis;  // -1: no usable plane, cut the range in the middle
    int32_t  split;
    float    scale;
    float    cmin;
    uint32_t nL;
    uint32_t node;
    uint32_t child;
    uint32_t pad;
};

PRT_DEV uint32_t fkey(float f)
{
    const uint32_t u = (uint32_t)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
PRT_DEV float fdec(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __int_as_float((int)u);
}
constexpr uint32_t kKeyPosInf = 0xFF800000u;  // fkey(+inf)
constexpr uint32_t kKeyNegInf = 0x007FFFFFu;  // fkey(-inf)

PRT_DEV int bin_of(float c, float cmin, float scale)
{
    int b = (int)((c - cmin) * scale);
    return b > kBins - 1 ? kBins - 1 : b;
}

// ---- exclusive scan (3 kernels) ---------------------------------------------------------------
#define PRT_SCAN_BLOCK 256
#define PRT_SCAN_ITEMS 4
#define PRT_SCAN_TILE (PRT_SCAN_BLOCK * PRT_SCAN_ITEMS)

PRT_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /*[PRT_SCAN_BLOCK/64]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t  inc  = v;
#pragma unroll
    for(int o = 1; o < 64; o <<= 1)
    {
        const uint32_t t = __shfl_up(inc, o, 64);
        if(lane >= o) inc += t;
    }
    if(lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for(int w = 0; w < PRT_SCAN_BLOCK / 64; ++w)
    {
        const uint32_t s = lds[w];
        if(w < wid) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + inc - v;
}

__global__ void __launch_bounds__(PRT_SCAN_BLOCK) k_scan_tiles(uint32_t* __restrict__ data, uint32_t n,
                                                              uint32_t* __restrict__ tileSums)
{
    __shared__ uint32_t lds[PRT_SCAN_BLOCK / 64];
    const uint32_t      base = blockIdx.x * PRT_SCAN_TILE + threadIdx.x * PRT_SCAN_ITEMS;
    uint32_t            v[PRT_SCAN_ITEMS], sum = 0;
#pragma unroll
    for(int k = 0; k < PRT_SCAN_ITEMS; ++k)
    {
        v[k] = (base + k < n) ? data[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, lds, total);
#pragma unroll
    for(int k = 0; k < PRT_SCAN_ITEMS; ++k)
    {
        if(base + k < n) data[base + k] = run;
        run += v[k];
    }
    if(threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(PRT_SCAN_BLOCK) k_scan_sums(uint32_t* __restrict__ tileSums, uint32_t ntiles,
                                                             uint32_t* __restrict__ totalOut)
{
    __shared__ uint32_t lds[PRT_SCAN_BLOCK / 64];
    uint32_t            carry = 0;
    for(uint32_t base = 0; base < ntiles; base += PRT_SCAN_BLOCK)
    {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? tileSums[i] : 0u;
        uint32_t       total;
        const uint32_t e = block_exclusive_scan(v, lds, total);
        if(i < ntiles) tileSums[i] = carry + e;
        carry += total;
    }
    if(threadIdx.x == 0) *totalOut = carry;
}
__global__ void __launch_bounds__(PRT_SCAN_BLOCK) k_scan_add(uint32_t* __restrict__ data, uint32_t n,
                                                            const uint32_t* __restrict__ tileSums)
{
    const uint32_t base = blockIdx.x * PRT_SCAN_TILE + threadIdx.x * PRT_SCAN_ITEMS;
    const uint32_t off  = tileSums[blockIdx.x];
#pragma unroll
    for(int k = 0; k < PRT_SCAN_ITEMS; ++k)
        if(base + k < n) data[base + k] += off;
}

// ---- build kernels ------------------------------------------------------------------------------
__global__ void k_build_reset(BuildNode* nodes, uint32_t* cb, uint32_t ntri)
{
    if(threadIdx.x == 0 && blockIdx.x == 0)
    {
        for(int k = 0; k < 3; ++k)
        {
            nodes[0].bmin[k] = kKeyPosInf;
            nodes[0].bmax[k] = kKeyNegInf;
            cb[k]            = kKeyPosInf;
            cb[3 + k]        = kKeyNegInf;
        }
        nodes[0].first = 0;
        nodes[0].count = ntri;
    }
}

// per-primitive box (triangles: exact min/max of the three vertices; `given` != nullptr: the boxes of the top level's
// instances, 2 x float4 each, as they are), identity order, root bounds
__global__ void __launch_bounds__(256) k_prim_boxes(const prt_vertex* __restrict__ verts,
                                                    const uint32_t* __restrict__ idx, uint32_t ntri,
                                                    float4* __restrict__ pbox, uint32_t* __restrict__ order,
                                                    uint32_t* __restrict__ nodeOf, BuildNode* nodes, uint32_t* cb,
                                                    const float4* __restrict__ given)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       kmn[3] = {kKeyPosInf, kKeyPosInf, kKeyPosInf}, kmx[3] = {kKeyNegInf, kKeyNegInf, kKeyNegInf};
    uint32_t       cmn[3] = {kKeyPosInf, kKeyPosInf, kKeyPosInf}, cmx[3] = {kKeyNegInf, kKeyNegInf, kKeyNegInf};
    if(i < ntri)
    {
        if(given)
        {
            const float4 lo = given[2 * i], hi = given[2 * i + 1];
            kmn[0] = fkey(lo.x); kmn[1] = fkey(lo.y); kmn[2] = fkey(lo.z);
            kmx[0] = fkey(hi.x); kmx[1] = fkey(hi.y); kmx[2] = fkey(hi.z);
        }
        else
            for(int k = 0; k < 3; ++k)
            {
                const float* p = verts[idx[3 * i + k]].p;
                for(int a = 0; a < 3; ++a)
                {
                    const uint32_t key = fkey(p[a]);
                    kmn[a]             = min(kmn[a], key);
                    kmx[a]             = max(kmx[a], key);
                }
            }
        const float mn[3] = {fdec(kmn[0]), fdec(kmn[1]), fdec(kmn[2])};
        const float mx[3] = {fdec(kmx[0]), fdec(kmx[1]), fdec(kmx[2])};
        pbox[2 * i]       = make_float4(mn[0], mn[1], mn[2], 0.0f);
        pbox[2 * i + 1]   = make_float4(mx[0], mx[1], mx[2], 0.0f);
        order[i]          = i;
        nodeOf[i]         = 0;
        for(int a = 0; a < 3; ++a)
        {
            const uint32_t ck = fkey((mn[a] + mx[a]) * 0.5f);
            cmn[a]            = ck;
            cmx[a]            = ck;
        }
    }
    // wave-level reduction, then one atomic per wave and bound
    for(int a = 0; a < 3; ++a)
    {
        for(int o = 32; o > 0; o >>= 1)
        {
            kmn[a] = min(kmn[a], (uint32_t)__shfl_xor((int)kmn[a], o, 64));
            kmx[a] = max(kmx[a], (uint32_t)__shfl_xor((int)kmx[a], o, 64));
            cmn[a] = min(cmn[a], (uint32_t)__shfl_xor((int)cmn[a], o, 64));
            cmx[a] = max(cmx[a], (uint32_t)__shfl_xor((int)cmx[a], o, 64));
        }
        if((threadIdx.x & 63) == 0)
        {
            atomicMin(&nodes[0].bmin[a], kmn[a]);
            atomicMax(&nodes[0].bmax[a], kmx[a]);
            atomicMin(&cb[a], cmn[a]);
            atomicMax(&cb[3 + a], cmx[a]);
        }
    }
}

// flags[j] = 1 when node levelStart+j must be split
// leafMax: PRT_LEAF_MAX for triangles; 1 for the top level of a two-level scene (one instance per leaf)
__global__ void __launch_bounds__(256) k_level_flags(const BuildNode* __restrict__ nodes, uint32_t levelStart,
                                                     uint32_t levelCount, uint32_t* __restrict__ flags, uint32_t leafMax)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j < levelCount) flags[j] = nodes[levelStart + j].count > leafMax ? 1u : 0u;
}

// after the scan: rank[j] valid for splitting nodes; record node id per split, clear the bins
__global__ void __launch_bounds__(256) k_level_prepare(const BuildNode* __restrict__ nodes, uint32_t levelStart,
                                                       uint32_t levelCount, const uint32_t* __restrict__ rank,
                                                       SplitRec* __restrict__ splits, uint32_t leafMax)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j < levelCount && nodes[levelStart + j].count > leafMax) splits[rank[j]].node = levelStart + j;
}
__global__ void __launch_bounds__(256) k_clear_bins(uint32_t* __restrict__ bins, uint32_t nsplit)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= (size_t)nsplit * kBinWords) return;
    const uint32_t w = (uint32_t)(i % kBinWords);
    bins[i]          = w < 3 * kBins ? 0u : (w < 3 * kBins * 4 ? kKeyPosInf : kKeyNegInf);
}

// every triangle of a splitting node drops its box into one bin per axis.
// Triangles are sorted by node, so on the upper levels a whole block belongs to ONE node: its 336 bin words are
// then accumulated in LDS and flushed with one global atomic per word (same-address global atomics serialise;
// they were 90 % of the build time). Mixed blocks (lower levels, little contention) go to global memory directly.
// Integer add / min / max commute, so the bins - and the tree - do not depend on the path taken.
__global__ void __launch_bounds__(256) k_bin(const uint32_t* __restrict__ order, const uint32_t* __restrict__ nodeOf,
                                             const BuildNode* __restrict__ nodes, const uint32_t* __restrict__ cb,
                                             const float4* __restrict__ pbox, const uint32_t* __restrict__ rank,
                                             uint32_t levelStart, uint32_t ntri, uint32_t* __restrict__ bins,
                                             uint32_t leafMax)
{
    __shared__ uint32_t lbins[kBinWords];
    const uint32_t      i     = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t      first = blockIdx.x * blockDim.x;
    const uint32_t      last  = min(first + blockDim.x, ntri) - 1u;
    const uint32_t      n0    = nodeOf[first];
    const bool          uniform = n0 == nodeOf[last];  // block-uniform: sorted by node
    if(uniform)
    {
        if(nodes[n0].count <= leafMax) return;  // the whole block sits in a finished leaf
        for(uint32_t w = threadIdx.x; w < kBinWords; w += blockDim.x)
            lbins[w] = w < 3 * kBins ? 0u : (w < 3 * kBins * 4 ? kKeyPosInf : kKeyNegInf);
        __syncthreads();
    }
    if(i < ntri)
    {
        const uint32_t ni = nodeOf[i];
        if(nodes[ni].count > leafMax)
        {
            const uint32_t p  = order[i];
            const float4   mn = pbox[2 * p], mx = pbox[2 * p + 1];
            const float    c[3]  = {(mn.x + mx.x) * 0.5f, (mn.y + mx.y) * 0.5f, (mn.z + mx.z) * 0.5f};
            const uint32_t km[3] = {fkey(mn.x), fkey(mn.y), fkey(mn.z)};
            const uint32_t kx[3] = {fkey(mx.x), fkey(mx.y), fkey(mx.z)};
            uint32_t*      B     = uniform ? lbins : bins + (size_t)rank[ni - levelStart] * kBinWords;
            for(int a = 0; a < 3; ++a)
            {
                const float cmin = fdec(cb[6 * ni + a]), cmax = fdec(cb[6 * ni + 3 + a]);
                const float ext = cmax - cmin;
                if(!(ext > 0.0f)) continue;
                const float scale = (float)kBins / ext;
                const int   b     = bin_of(c[a], cmin, scale);
                atomicAdd(&B[a * kBins + b], 1u);
                uint32_t* bmn = B + 3 * kBins + (a * kBins + b) * 3;
                uint32_t* bmx = B + 3 * kBins * 4 + (a * kBins + b) * 3;
                for(int k = 0; k < 3; ++k)
                {
                    atomicMin(&bmn[k], km[k]);
                    atomicMax(&bmx[k], kx[k]);
                }
            }
        }
    }
    if(uniform)
    {
        __syncthreads();
        uint32_t* B = bins + (size_t)rank[n0 - levelStart] * kBinWords;
        for(uint32_t w = threadIdx.x; w < kBinWords; w += blockDim.x)
        {
            const uint32_t v = lbins[w];
            if(w < 3 * kBins)
            {
                if(v) atomicAdd(&B[w], v);
            }
            else if(w < 3 * kBins * 4)
            {
                if(v != kKeyPosInf) atomicMin(&B[w], v);
            }
            else if(v != kKeyNegInf)
                atomicMax(&B[w], v);
        }
    }
}

struct KBox
{
    uint32_t mn[3], mx[3];
};
PRT_DEV void kbox_empty(KBox& b)
{
    for(int k = 0; k < 3; ++k) { b.mn[k] = kKeyPosInf; b.mx[k] = kKeyNegInf; }
}
PRT_DEV float kbox_area(const KBox& b)
{
    const float dx = fdec(b.mx[0]) - fdec(b.mn[0]), dy = fdec(b.mx[1]) - fdec(b.mn[1]),
                dz = fdec(b.mx[2]) - fdec(b.mn[2]);
    return ((dx * dy) + (dy * dz)) + (dz * dx);
}

// one thread per splitting node: sweep 3 x 15 planes, keep the cheapest (first wins ties), emit children
__global__ void __launch_bounds__(64) k_split(BuildNode* __restrict__ nodes, uint32_t* __restrict__ cb,
                                              const uint32_t* __restrict__ bins, SplitRec* __restrict__ splits,
                                              uint32_t nsplit, uint32_t childBase)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nsplit) return;
    const uint32_t  ni    = splits[s].node;
    const uint32_t  start = nodes[ni].first, count = nodes[ni].count;
    const uint32_t* B     = bins + (size_t)s * kBinWords;
    float           bestCost = __builtin_inff();
    int             bestAxis = -1, bestSplit = 0;
    float           bestScale = 0.0f, bestCmin = 0.0f;
    uint32_t        bestNL = 0;
    KBox            bestL, bestR;
    kbox_empty(bestL);
    kbox_empty(bestR);
    for(int a = 0; a < 3; ++a)
    {
        const float cmin = fdec(cb[6 * ni + a]), cmax = fdec(cb[6 * ni + 3 + a]);
        const float ext = cmax - cmin;
        if(!(ext > 0.0f)) continue;
        const float     scale = (float)kBins / ext;
        const uint32_t* cnt   = B + a * kBins;
        const uint32_t* bmn   = B + 3 * kBins + a * kBins * 3;
        const uint32_t* bmx   = B + 3 * kBins * 4 + a * kBins * 3;
        // suffix areas / counts
        float    areaR[kBins];
        uint32_t nRs[kBins];
        KBox     acc;
        kbox_empty(acc);
        uint32_t nacc = 0;
        for(int b = kBins - 1; b >= 1; --b)
        {
            for(int k = 0; k < 3; ++k)
            {
                acc.mn[k] = min(acc.mn[k], bmn[b * 3 + k]);
                acc.mx[k] = max(acc.mx[k], bmx[b * 3 + k]);
            }
            nacc += cnt[b];
            nRs[b]   = nacc;
            areaR[b] = nacc ? kbox_area(acc) : 0.0f;
        }
        KBox L;
        kbox_empty(L);
        uint32_t nL = 0;
        for(int sp = 1; sp < kBins; ++sp)
        {
            for(int k = 0; k < 3; ++k)
            {
                L.mn[k] = min(L.mn[k], bmn[(sp - 1) * 3 + k]);
                L.mx[k] = max(L.mx[k], bmx[(sp - 1) * 3 + k]);
            }
            nL += cnt[sp - 1];
            const uint32_t nR = nRs[sp];
            if(nL == 0 || nR == 0) continue;
            const float cost = (kbox_area(L) * (float)nL) + (areaR[sp] * (float)nR);
            if(cost < bestCost)
            {
                bestCost  = cost;
                bestAxis  = a;
                bestSplit = sp;
                bestScale = scale;
                bestCmin  = cmin;
                bestNL    = nL;
                bestL     = L;
            }
        }
    }
    if(bestAxis >= 0)
    {
        // right box of the winning plane
        const uint32_t* bmn = B + 3 * kBins + bestAxis * kBins * 3;
        const uint32_t* bmx = B + 3 * kBins * 4 + bestAxis * kBins * 3;
        for(int b = bestSplit; b < kBins; ++b)
            for(int k = 0; k < 3; ++k)
            {
                bestR.mn[k] = min(bestR.mn[k], bmn[b * 3 + k]);
                bestR.mx[k] = max(bestR.mx[k], bmx[b * 3 + k]);
            }
    }
    else
    {
        bestNL = count / 2;  // boxes are accumulated by k_scatter
    }
    const uint32_t child = childBase + 2 * s;
    SplitRec       r;
    r.axis  = bestAxis;
    r.split = bestSplit;
    r.scale = bestScale;
    r.cmin  = bestCmin;
    r.nL    = bestNL;
    r.node  = ni;
    r.child = child;
    r.pad   = 0;
    splits[s] = r;
    BuildNode l, rr;
    for(int k = 0; k < 3; ++k)
    {
        l.bmin[k]  = bestL.mn[k];
        l.bmax[k]  = bestL.mx[k];
        rr.bmin[k] = bestR.mn[k];
        rr.bmax[k] = bestR.mx[k];
    }
    l.first  = start;
    l.count  = bestNL;
    rr.first = start + bestNL;
    rr.count = count - bestNL;
    nodes[child]     = l;
    nodes[child + 1] = rr;
    nodes[ni].first  = child;
    nodes[ni].count  = 0;
    for(int k = 0; k < 3; ++k)
    {
        cb[6 * child + k]           = kKeyPosInf;
        cb[6 * child + 3 + k]       = kKeyNegInf;
        cb[6 * (child + 1) + k]     = kKeyPosInf;
        cb[6 * (child + 1) + 3 + k] = kKeyNegInf;
    }
}

// NOTE: k_flags / k_scatter run after k_split rewrote nodes[ni] (count = 0, first = child), so they
// identify splitting nodes through `splitOf` (rank per level node, ~0u for leaves) captured before.
__global__ void __launch_bounds__(256) k_split_of(const BuildNode* __restrict__ nodes, uint32_t levelStart,
                                                  uint32_t levelCount, const uint32_t* __restrict__ rank,
                                                  uint32_t* __restrict__ splitOf, uint32_t leafMax)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j < levelCount) splitOf[j] = nodes[levelStart + j].count > leafMax ? rank[j] : 0xffffffffu;
}

__global__ void __launch_bounds__(256) k_flags(const uint32_t* __restrict__ order, const uint32_t* __restrict__ nodeOf,
                                               const uint32_t* __restrict__ splitOf, const SplitRec* __restrict__ splits,
                                               const BuildNode* __restrict__ nodes, const float4* __restrict__ pbox,
                                               uint32_t levelStart, uint32_t ntri, uint32_t* __restrict__ flags,
                                               uint32_t* __restrict__ scanBuf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= ntri) return;
    const uint32_t ni = nodeOf[i];
    uint32_t       f  = 0;
    if(ni >= levelStart)
    {
        const uint32_t s = splitOf[ni - levelStart];
        if(s != 0xffffffffu)
        {
            const SplitRec r = splits[s];
            if(r.axis >= 0)
            {
                const uint32_t p  = order[i];
                const float4   mn = pbox[2 * p], mx = pbox[2 * p + 1];
                const float    c  = r.axis == 0 ? (mn.x + mx.x) * 0.5f
                                                : (r.axis == 1 ? (mn.y + mx.y) * 0.5f : (mn.z + mx.z) * 0.5f);
                f = bin_of(c, r.cmin, r.scale) < r.split ? 1u : 0u;
            }
            else
            {
                // median cut: the first nL positions of the range (range start = left child's first)
                f = (i - nodes[r.child].first) < r.nL ? 1u : 0u;
            }
        }
    }
    flags[i]   = f;
    scanBuf[i] = f;
}
__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ order, const uint32_t* __restrict__ nodeOf,
                                                 const uint32_t* __restrict__ splitOf,
                                                 const SplitRec* __restrict__ splits, BuildNode* nodes,
                                                 uint32_t* cb, const float4* __restrict__ pbox,
                                                 const uint32_t* __restrict__ flagsIn,
                                                 const uint32_t* __restrict__ scanL, uint32_t levelStart,
                                                 uint32_t ntri, uint32_t* __restrict__ orderOut,
                                                 uint32_t* __restrict__ nodeOfOut)
{
    // [child 0/1][centroid min 3 | centroid max 3 | box min 3 | box max 3]: block-private when the whole block
    // belongs to one splitting node (see k_bin)
    __shared__ uint32_t lcb[2][12];
    const uint32_t      i     = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t      first = blockIdx.x * blockDim.x;
    const uint32_t      last  = min(first + blockDim.x, ntri) - 1u;
    const uint32_t      n0    = nodeOf[first];
    const bool          uniform = n0 == nodeOf[last];
    if(uniform)
    {
        if(threadIdx.x < 24) lcb[threadIdx.x / 12][threadIdx.x % 12] = ((threadIdx.x % 6) < 3) ? kKeyPosInf : kKeyNegInf;
        __syncthreads();
    }
    uint32_t childBase = 0xffffffffu;
    bool     median    = false;
    if(i < ntri)
    {
        const uint32_t ni = nodeOf[i];
        const uint32_t p  = order[i];
        uint32_t       s  = 0xffffffffu;
        if(ni >= levelStart) s = splitOf[ni - levelStart];
        if(s == 0xffffffffu)
        {
            orderOut[i]  = p;
            nodeOfOut[i] = ni;
        }
        else
        {
            const SplitRec r     = splits[s];
            const uint32_t start = nodes[r.child].first;
            const uint32_t lBefore = scanL[i] - scanL[start];
            const bool     left    = flagsIn[i] != 0u;
            const uint32_t pos     = left ? start + lBefore : start + r.nL + ((i - start) - lBefore);
            const uint32_t child   = left ? r.child : r.child + 1;
            orderOut[pos]          = p;
            nodeOfOut[pos]         = child;
            childBase              = r.child;
            median                 = r.axis < 0;
            const float4 mn = pbox[2 * p], mx = pbox[2 * p + 1];
            const float  c[3] = {(mn.x + mx.x) * 0.5f, (mn.y + mx.y) * 0.5f, (mn.z + mx.z) * 0.5f};
            uint32_t*    ccb  = uniform ? &lcb[left ? 0 : 1][0] : cb + 6 * (size_t)child;
            for(int a = 0; a < 3; ++a)
            {
                const uint32_t ck = fkey(c[a]);
                atomicMin(&ccb[a], ck);
                atomicMax(&ccb[3 + a], ck);
            }
            if(median)
            {
                const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
                for(int a = 0; a < 3; ++a)
                {
                    if(uniform)
                    {
                        atomicMin(&lcb[left ? 0 : 1][6 + a], fkey(lo[a]));
                        atomicMax(&lcb[left ? 0 : 1][9 + a], fkey(hi[a]));
                    }
                    else
                    {
                        atomicMin(&nodes[child].bmin[a], fkey(lo[a]));
                        atomicMax(&nodes[child].bmax[a], fkey(hi[a]));
                    }
                }
            }
        }
    }
    if(uniform)
    {
        // childBase / median are the same for every triangle of the block (one node): take them from any lane
        __shared__ uint32_t sChild, sMedian;
        if(threadIdx.x == 0) { sChild = 0xffffffffu; sMedian = 0u; }
        __syncthreads();
        if(childBase != 0xffffffffu) { sChild = childBase; sMedian = median ? 1u : 0u; }
        __syncthreads();
        if(sChild != 0xffffffffu && threadIdx.x < 24)
        {
            const uint32_t ch = sChild + threadIdx.x / 12, w = threadIdx.x % 12;
            const uint32_t v  = lcb[threadIdx.x / 12][w];
            const bool     isMin = (w % 6) < 3;
            if(v != (isMin ? kKeyPosInf : kKeyNegInf))
            {
                if(w < 6)
                {
                    if(isMin) atomicMin(&cb[6 * (size_t)ch + w], v);
                    else atomicMax(&cb[6 * (size_t)ch + w], v);
                }
                else if(sMedian)
                {
                    if(isMin) atomicMin(&nodes[ch].bmin[w - 6], v);
                    else atomicMax(&nodes[ch].bmax[w - 9], v);
                }
            }
        }
    }
}

// ---- emit the traversal layout: collapse the binary build tree into BVH4 ------------------------------
// A wide node is made from every inner build node on an EVEN level: its slots are its grandchildren
// (children that are leaves stay as they are), in left-to-right order. Child reference: >= 0 index of a
// wide node, < 0 ~((first << 3) | count) for a leaf over tris[first .. first+count); empty slots carry an
// inverted box that no ray enters. Boxes are stored as 16-bit fixed point on the root box, rounded outwards.
struct LevelTable
{
    uint32_t start[PRT_MAX_DEPTH + 2];  // first build-node index of every level (levels are contiguous)
    uint32_t n;
};
PRT_DEV uint32_t level_of(const LevelTable& lt, uint32_t i)
{
    uint32_t l = 0;
    while(l + 1 < lt.n && i >= lt.start[l + 1]) ++l;
    return l;
}
__global__ void __launch_bounds__(256) k_wide_flags(const BuildNode* __restrict__ nodes, uint32_t nnodes,
                                                    LevelTable lt, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < nnodes) flags[i] = (nodes[i].count == 0 && (level_of(lt, i) & 1u) == 0u) ? 1u : 0u;
}
struct QuantGrid
{
    float qmin[3], qscale[3];  // coordinate = qmin + q * qscale
};
// largest q with qmin + q*scale <= v / smallest q with qmin + q*scale >= v, exact in fp64
// (q*scale and the sum are exact in double: 16 x 24 bits, then < 53 bits)
PRT_DEV uint32_t quant_floor(float v, float qmin, float scale)
{
    double q = __builtin_floor(((double)v - (double)qmin) / (double)scale);
    if(q < 0.0) q = 0.0;
    if(q > 65535.0) q = 65535.0;
    while(q > 0.0 && (double)qmin + q * (double)scale > (double)v) q -= 1.0;
    return (uint32_t)q;
}
PRT_DEV uint32_t quant_ceil(float v, float qmin, float scale)
{
    double q = __builtin_ceil(((double)v - (double)qmin) / (double)scale);
    if(q < 0.0) q = 0.0;
    if(q > 65535.0) q = 65535.0;
    while(q < 65535.0 && (double)qmin + q * (double)scale < (double)v) q += 1.0;
    return (uint32_t)q;
}
// Position of this structure in the node / leaf arrays that the structures of a two-level scene share (single-level:
// all zero); child references are absolute. The first hotCount nodes of the tree (its top levels: nodes are numbered
// level by level) go to hotBase.., the others to coldBase..: the hot parts of all structures form the front of the node
// array, which the traversal kernels stage in LDS (BvhView::nTop).
struct NodePlace
{
    uint32_t hotBase, hotCount, coldBase, slotBase;
};
PRT_DEV uint32_t node_place(const NodePlace& p, uint32_t w) { return w < p.hotCount ? p.hotBase + w : p.coldBase + (w - p.hotCount); }
// The code of a leaf reference (prt_trace.hpp, leaf_test): its triangle count, or 3 for two triangles that form a fan
// (A, B, C), (A, C, D) — same positions bit for bit — and can share one 64-byte slot. `swap`: the leaf's second triangle
// is the one that comes first in the fan. verts == nullptr (the top level of a two-level scene: leaves are instances) or
// a tree that is one leaf: plain counts.
PRT_DEV bool same_pos(const float* a, const float* b)
{
    return __float_as_uint(a[0]) == __float_as_uint(b[0]) && __float_as_uint(a[1]) == __float_as_uint(b[1]) &&
           __float_as_uint(a[2]) == __float_as_uint(b[2]);
}
PRT_DEV bool is_fan(const prt_vertex* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t p, uint32_t q)
{
    return same_pos(verts[idx[3 * p]].p, verts[idx[3 * q]].p) && same_pos(verts[idx[3 * p + 2]].p, verts[idx[3 * q + 1]].p);
}
PRT_DEV uint32_t leaf_code(const prt_vertex* __restrict__ verts, const uint32_t* __restrict__ idx,
                           const uint32_t* __restrict__ order, uint32_t first, uint32_t count, bool& swap)
{
    swap = false;
#if PRT_LEAF_QUAD
    if(verts && count == 2u)
    {
        const uint32_t p = order[first], q = order[first + 1];
        if(is_fan(verts, idx, p, q)) return 3u;
        if(is_fan(verts, idx, q, p))
        {
            swap = true;
            return 3u;
        }
    }
#endif
    return count;
}
__global__ void __launch_bounds__(256) k_emit_wide(const BuildNode* __restrict__ nodes, uint32_t nnodes,
                                                   LevelTable lt, const uint32_t* __restrict__ wideIdx, QuantGrid g,
                                                   uint4* __restrict__ out, NodePlace pl,
                                                   const prt_vertex* __restrict__ verts, const uint32_t* __restrict__ idx,
                                                   const uint32_t* __restrict__ order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes || nodes[i].count != 0 || (level_of(lt, i) & 1u) != 0u) return;
    uint32_t slot[4];
    int      tri[4] = {-1, -1, -1, -1};  // >= 0: this triangle of the leaf alone (a split pair, below)
    int      n = 0;
    for(uint32_t c = nodes[i].first; c < nodes[i].first + 2; ++c)
    {
        if(nodes[c].count == 0)
        {
            slot[n++] = nodes[c].first;
            slot[n++] = nodes[c].first + 1;
        }
        else
            slot[n++] = c;
    }
#if PRT_LEAF_SPLIT_PAIR
    // A free child slot takes one half of a pair of triangles that is no fan: two single-triangle leaves with their own
    // boxes instead of one leaf that needs two fetches (every triangle has its own 64-byte slot, k_emit_leaves).
    if(verts)
        for(int j = 0, n0 = n; j < n0 && n < 4; ++j)
        {
            const BuildNode c = nodes[slot[j]];
            bool            swap;
            if(c.count == 2u && leaf_code(verts, idx, order, c.first, c.count, swap) == 2u)
            {
                tri[j]    = 0;
                slot[n]   = slot[j];
                tri[n++]  = 1;
            }
        }
#endif
    uint32_t qmn[3][4], qmx[3][4];
    int32_t  ref[4];
    for(int j = 0; j < 4; ++j)
    {
        if(j < n)
        {
            const BuildNode c = nodes[slot[j]];
            float           lo[3], hi[3];
            for(int k = 0; k < 3; ++k)
            {
                lo[k] = fdec(c.bmin[k]);
                hi[k] = fdec(c.bmax[k]);
            }
            if(tri[j] >= 0)
            {
                const uint32_t p = order[c.first + (uint32_t)tri[j]];
                for(int k = 0; k < 3; ++k)
                {
                    const float a = verts[idx[3 * p]].p[k], b = verts[idx[3 * p + 1]].p[k], d = verts[idx[3 * p + 2]].p[k];
                    lo[k]         = fminf(a, fminf(b, d));
                    hi[k]         = fmaxf(a, fmaxf(b, d));
                }
            }
            for(int k = 0; k < 3; ++k)
            {
                qmn[k][j] = quant_floor(lo[k], g.qmin[k], g.qscale[k]);
                qmx[k][j] = quant_ceil(hi[k], g.qmin[k], g.qscale[k]);
#ifdef PRT_QUANT_COARSE  // measurement builds: what boxes on a 2^(16 - PRT_QUANT_COARSE) grid cost in node visits
                qmn[k][j] = (qmn[k][j] >> PRT_QUANT_COARSE) << PRT_QUANT_COARSE;
                qmx[k][j] = ((qmx[k][j] + (1u << PRT_QUANT_COARSE) - 1u) >> PRT_QUANT_COARSE) << PRT_QUANT_COARSE;
                if(qmx[k][j] > 65535u) qmx[k][j] = 65535u;
#endif
            }
            bool swap;
            if(tri[j] >= 0)
                ref[j] = (int32_t) ~(((c.first + (uint32_t)tri[j] + pl.slotBase) << 3) | 1u);
            else
                ref[j] = c.count == 0 ? (int32_t)node_place(pl, wideIdx[slot[j]])
                                      : (int32_t)~(((c.first + pl.slotBase) << 3) | leaf_code(verts, idx, order, c.first, c.count, swap));
        }
        else
        {
            // empty slot: inverted box; should a ray still "enter" it (flat scene), the reference is an empty leaf
            for(int k = 0; k < 3; ++k)
            {
                qmn[k][j] = 65535u;
                qmx[k][j] = 0u;
            }
            ref[j] = -1;
        }
    }
    uint4* o = out + (size_t)node_place(pl, wideIdx[i]) * 4;
    for(int k = 0; k < 3; ++k)
        o[k] = make_uint4(qmn[k][0] | (qmn[k][1] << 16), qmn[k][2] | (qmn[k][3] << 16), qmx[k][0] | (qmx[k][1] << 16),
                          qmx[k][2] | (qmx[k][3] << 16));
    o[3] = make_uint4((uint32_t)ref[0], (uint32_t)ref[1], (uint32_t)ref[2], (uint32_t)ref[3]);
}
// Leaf records. A leaf over order[first .. first+count) owns the 64-byte slots first .. first+count-1 of the triangle
// array and keeps its triangles packed from the start of slot `first`, 3 x float4 each (prim id in v0.w) — or, for a
// fan (leaf_code 3), its four corners in slot `first` alone: (A | id of the first triangle), (B | id of the second),
// (C | 0), (D | 0). Measured (tools/ubench_gather.hip): what a divergent gather costs a CU is the number of 64-byte
// sectors it touches; a single triangle and a fan touch one. `fans` = false for a tree that is one leaf (its reference
// is made on the host from the count alone).
__global__ void __launch_bounds__(256) k_emit_leaves(const BuildNode* __restrict__ nodes, uint32_t nnodes,
                                                     const prt_vertex* __restrict__ verts,
                                                     const uint32_t* __restrict__ idx,
                                                     const uint32_t* __restrict__ order, float4* __restrict__ tris,
                                                     uint32_t slotBase, uint32_t fans)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes) return;
    const uint32_t first = nodes[i].first, count = nodes[i].count;
    if(count == 0) return;
    float4* o = tris + (size_t)(first + slotBase) * PRT_LEAF_SLOT_F4;
    bool    swap = false;
    if(fans && leaf_code(verts, idx, order, first, count, swap) == 3u)
    {
        const uint32_t p = order[first + (swap ? 1u : 0u)], q = order[first + (swap ? 0u : 1u)];
        const float*   a = verts[idx[3 * p]].p;
        const float*   b = verts[idx[3 * p + 1]].p;
        const float*   c = verts[idx[3 * p + 2]].p;
        const float*   d = verts[idx[3 * q + 2]].p;
        o[0]             = make_float4(a[0], a[1], a[2], __int_as_float((int)p));
        o[1]             = make_float4(b[0], b[1], b[2], __int_as_float((int)q));
        o[2]             = make_float4(c[0], c[1], c[2], 0.0f);
        o[3]             = make_float4(d[0], d[1], d[2], 0.0f);
        return;
    }
    for(uint32_t q = 0; q < count; ++q)
    {
        const uint32_t p = order[first + q];
        const float*   a = verts[idx[3 * p]].p;
        const float*   b = verts[idx[3 * p + 1]].p;
        const float*   c = verts[idx[3 * p + 2]].p;
        float4*        w = o + (PRT_LEAF_SPLIT_PAIR ? PRT_LEAF_SLOT_F4 : 3) * q;  // split pairs: one triangle per slot
        w[0]             = make_float4(a[0], a[1], a[2], __int_as_float((int)p));
        w[1]             = make_float4(b[0], b[1], b[2], 0.0f);
        w[2]             = make_float4(c[0], c[1], c[2], 0.0f);
    }
}
// float view of the build nodes for prt_read_bvh()
__global__ void __launch_bounds__(256) k_decode_nodes(const BuildNode* __restrict__ nodes, uint32_t nnodes,
                                                      float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes) return;
    const BuildNode n = nodes[i];
    out[2 * i]     = make_float4(fdec(n.bmin[0]), fdec(n.bmin[1]), fdec(n.bmin[2]), __int_as_float((int)n.first));
    out[2 * i + 1] = make_float4(fdec(n.bmax[0]), fdec(n.bmax[1]), fdec(n.bmax[2]), __int_as_float((int)n.count));
}

// ---- refit (VkAccelerationStructure "updateOnly": vkRT_BLAS.cpp:209-290 / vkRT_TLAS.cpp:178-279 with
// updateOnly = true): the topology and the triangle order of the last build stay, the boxes follow the moved
// vertices. Leaves first, then one launch per level from the deepest one up (levels are contiguous).
__global__ void __launch_bounds__(256) k_refit_leaves(BuildNode* __restrict__ nodes, uint32_t nnodes,
                                                      const prt_vertex* __restrict__ verts,
                                                      const uint32_t* __restrict__ idx,
                                                      const uint32_t* __restrict__ order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes) return;
    const uint32_t first = nodes[i].first, count = nodes[i].count;
    if(count == 0) return;
    uint32_t kmn[3] = {kKeyPosInf, kKeyPosInf, kKeyPosInf}, kmx[3] = {kKeyNegInf, kKeyNegInf, kKeyNegInf};
    for(uint32_t t = first; t < first + count; ++t)
    {
        const uint32_t p = order[t];
        for(int k = 0; k < 3; ++k)
        {
            const float* q = verts[idx[3 * p + k]].p;
            for(int a = 0; a < 3; ++a)
            {
                const uint32_t key = fkey(q[a]);
                kmn[a]             = min(kmn[a], key);
                kmx[a]             = max(kmx[a], key);
            }
        }
    }
    for(int a = 0; a < 3; ++a)
    {
        nodes[i].bmin[a] = kmn[a];
        nodes[i].bmax[a] = kmx[a];
    }
}
__global__ void __launch_bounds__(256) k_refit_level(BuildNode* __restrict__ nodes, uint32_t levelStart,
                                                     uint32_t levelCount)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= levelCount) return;
    const uint32_t i = levelStart + j;
    if(nodes[i].count != 0) return;
    const uint32_t c = nodes[i].first;
    for(int a = 0; a < 3; ++a)
    {
        nodes[i].bmin[a] = min(nodes[c].bmin[a], nodes[c + 1].bmin[a]);
        nodes[i].bmax[a] = max(nodes[c].bmax[a], nodes[c + 1].bmax[a]);
    }
}

}  // namespace prt
