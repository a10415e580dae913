// oracle.cpp — TEST INFRASTRUCTURE, not product code (see oracle.h).
//
// Scalar CPU restatement of the reference's ray-tracing hot path. One C++ function per GLSL function,
// each citing the reference file:line it follows. Built with -ffp-contract=off so that every fp32
// expression is evaluated exactly as written (one IEEE-754 rounding per operator, in source order).
//
// PARITY UNPINNED for everything the reference leaves to the NVIDIA driver / GLSL built-ins:
//   * BVH build + traversal + ray/triangle test (vkCmdBuildAccelerationStructureNV / traceNV),
//   * sin/cos rounding, pow() with constant exponents, texture filtering weights,
//   * the mt19937/random_device scramble values (non-deterministic in the reference).
// The Sobol sampler IS pinned (tests/golden/sobol_*).
//
// Evaluation-order conventions used everywhere (they are part of the definition):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   mat3*vec3     = (c0*x + c1*y) + c2*z                 (column major, per component)
//   mat4*vec4     = (c0*x + c1*y) + (c2*z + c3*w)
//   normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
//   min/max/clamp = GLSL definitions: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x
//   pow(x,5)      = ((x*x)*(x*x))*x ; pow(x,4) = (x*x)*(x*x) ; pow(x,3) = (x*x)*x ; pow(x,2) = x*x
#include "oracle.h"

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// tiny vector algebra (GLSL semantics)
// ------------------------------------------------------------------------------------------------
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat3 { vec3 c0, c1, c2; };  // columns

inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 V3(float s) { return vec3{s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, vec3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator*(vec3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(vec3 a) { return V3(-a.x, -a.y, -a.z); }
inline vec2 operator+(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return vec2{a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return vec2{a.x * s, a.y * s}; }

inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline float gabs(float x) { return std::fabs(x); }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3  cross(vec3 a, vec3 b)
{
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline vec3  normalize(vec3 a)
{
    const float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline vec3 mul(const mat3& m, vec3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
// transpose(m) * v : row r of the transposed matrix is column r of m
inline vec3 mulT(const mat3& m, vec3 v) { return V3(dot(m.c0, v), dot(m.c1, v), dot(m.c2, v)); }
inline vec4 mul4(const float* m, vec4 v)  // column-major 4x4
{
    vec4 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * v.w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * v.w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * v.w);
    r.w = (m[3] * v.x + m[7] * v.y) + (m[11] * v.z + m[15] * v.w);
    return r;
}
inline float maxcoord(vec3 v) { return gmax(gmax(v.x, v.y), v.z); }  // pathRT.rgen:211-214

const float kPi = 3.141592653589f;  // pathRT.rgen:85 (rounds to float(pi))

// sin/cos: GLSL leaves their rounding open; this fixed polynomial (Cody-Waite reduction by pi/2,
// cephes single-precision minimax kernels) IS the definition shared with the HIP kernels.
inline void sincos_spec(float x, float* s_out, float* c_out)
{
    const float q = floorf(x * 0.636619772f + 0.5f);
    const int   k = (int)q;
    float       r = x - q * 1.5703125f;
    r             = r - q * 4.837512969970703125e-4f;
    r             = r - q * 7.54978995489188216e-8f;
    const float z = r * r;
    float       t = -1.9515295891e-4f * z;
    t             = t + 8.3321608736e-3f;
    t             = t * z;
    t             = t - 1.6666654611e-1f;
    t             = t * z;
    t             = t * r;
    const float s = t + r;
    t             = 2.443315711809948e-5f * z;
    t             = t - 1.388731625493765e-3f;
    t             = t * z;
    t             = t + 4.166664568298827e-2f;
    t             = t * z;
    t             = t * z;
    const float c = (t - 0.5f * z) + 1.0f;
    switch(k & 3)
    {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
struct Tex
{
    std::vector<uint8_t> rgba;
    uint32_t             w = 0, h = 0;
};

struct Box
{
    float mn[3], mx[3];
};

}  // namespace

// One acceleration structure over a range of the scene's triangles (the whole scene, or one bottom-level structure of
// an instanced scene). Primitive ids inside it are local to the range.
struct Geo
{
    uint32_t              firstTri = 0, nTri = 0;
    std::vector<orc_node> nodes;
    std::vector<uint32_t> order;
    uint32_t              depth    = 0;
    float                 sceneAbs = 0.0f;
};
// One instance of a bottom-level structure (vkRT_TLAS.cpp:58-72 AddInstance; VkGeometryInstance keeps the 3 x 4
// affine part of the transform, :212-214).
struct Inst
{
    float    o2w[16];   // object -> world, column-major
    float    w2o[16];   // its inverse (glm's cofactor formula, the same fp32 operation order)
    float    nrm[16];   // transpose(inverse): normals
    bool     identity = false;
    uint32_t blas = 0, primBase = 0, instanceId = 0;
    Box      world;     // conservative world-space box (only used to skip instances; never decides a hit)
};

struct orc_scene
{
    std::vector<prt_vertex>   verts;
    std::vector<uint32_t>     idx;
    std::vector<prt_material> mats;
    std::vector<Tex>          tex;
    std::vector<uint32_t>     sobol;
    uint32_t                  sobolDims = 0, sobolSize = 52;
    uint32_t                  seed      = 1;
    const uint32_t*           scrTable  = nullptr;
    uint32_t                  scrW = 0, scrH = 0;
    Geo                       geo;   // single-level: the whole scene
    std::vector<Geo>          blas;  // two-level: bottom-level structures ...
    std::vector<Inst>         inst;  // ... and their instances (empty = single-level)
    size_t                    ntri() const { return idx.size() / 3; }
};

namespace {

// ------------------------------------------------------------------------------------------------
// Sobol sampler — pathRT.rgen:220-235, AO.rgen:131-146, sobol.h:40-55
// ------------------------------------------------------------------------------------------------
inline float sobol1DSample(const orc_scene& sc, uint32_t index, uint32_t dimension, uint32_t scramble)
{
    uint32_t result = scramble;
    for(uint32_t i = dimension * sc.sobolSize; index != 0; index >>= 1, ++i)
    {
        if((index & 1u) == 1u)
            result ^= sc.sobol[i];
    }
    // `result * 2.3283064365386963e-10` : uint -> float (round to nearest even), times 2^-32
    return (float)result * 2.3283064365386963e-10f;
}

// pathRT.rgen:600-606
inline vec2 nextSquareSample(const orc_scene& sc, uint32_t index, uint32_t& dim, const uint32_t scramble[2])
{
    vec2 s;
    s.x = sobol1DSample(sc, index, dim++, scramble[0]);
    s.y = sobol1DSample(sc, index, dim++, scramble[1]);
    return s;
}

// Replacement for the reference's mt19937(random_device) scramble layers
// (vkRTX_setup.cpp:808-832): stateless, seeded, reproducible. 32 layers per pixel.
inline uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t scrambleValue(const orc_scene& sc, uint32_t px, uint32_t py, uint32_t w, uint32_t h, uint32_t layer)
{
    const uint32_t pixel = py * w + px;
    if(sc.scrTable)
        return sc.scrTable[(size_t)layer * ((size_t)w * h) + pixel];
    uint32_t x = lowbias32(pixel * 32u + layer + 0x9E3779B9u * sc.seed);
    return lowbias32(x ^ (sc.seed + 0x85EBCA6Bu));
}

// ------------------------------------------------------------------------------------------------
// BVH: binned SAH, breadth-first, deterministic (stands in for vkCmdBuildAccelerationStructureNV,
// vkRT_BLAS.cpp:271 — PARITY UNPINNED). 16 bins, leaves <= 2 triangles.
// ------------------------------------------------------------------------------------------------
const int      NB       = 16;
const uint32_t LEAF_MAX = 2;

// total order on floats used for box min/max (distinguishes -0 < +0 like the GPU's integer atomics)
inline uint32_t fkey(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float kmin(float a, float b) { return fkey(b) < fkey(a) ? b : a; }
inline float kmax(float a, float b) { return fkey(b) > fkey(a) ? b : a; }

inline void boxEmpty(Box& b)
{
    for(int k = 0; k < 3; ++k)
    {
        b.mn[k] = INFINITY;
        b.mx[k] = -INFINITY;
    }
}
inline void boxGrow(Box& b, const Box& o)
{
    for(int k = 0; k < 3; ++k)
    {
        b.mn[k] = kmin(b.mn[k], o.mn[k]);
        b.mx[k] = kmax(b.mx[k], o.mx[k]);
    }
}
inline float boxArea(const Box& b)
{
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return ((dx * dy) + (dy * dz)) + (dz * dx);
}

int buildBVH(const orc_scene& scene, Geo& sc)
{
    const uint32_t N = sc.nTri;
    sc.nodes.clear();
    sc.order.resize(N);
    if(N == 0)
        return -1;
    std::vector<Box>   pb(N);
    std::vector<float> cen((size_t)N * 3);
    for(uint32_t i = 0; i < N; ++i)
    {
        Box b;
        boxEmpty(b);
        for(int k = 0; k < 3; ++k)
        {
            const float* p = scene.verts[scene.idx[3 * (size_t)(sc.firstTri + i) + k]].p;
            for(int a = 0; a < 3; ++a)
            {
                b.mn[a] = kmin(b.mn[a], p[a]);
                b.mx[a] = kmax(b.mx[a], p[a]);
            }
        }
        pb[i] = b;
        for(int a = 0; a < 3; ++a)
            cen[(size_t)i * 3 + a] = (b.mn[a] + b.mx[a]) * 0.5f;
        sc.order[i] = i;
    }
    Box root;
    boxEmpty(root);
    for(uint32_t i = 0; i < N; ++i)
        boxGrow(root, pb[i]);
    orc_node n0;
    memcpy(n0.bmin, root.mn, 12);
    memcpy(n0.bmax, root.mx, 12);
    n0.first = 0;
    n0.count = N;
    sc.nodes.push_back(n0);
    sc.sceneAbs = 0.0f;
    for(int a = 0; a < 3; ++a)
    {
        sc.sceneAbs = gmax(sc.sceneAbs, gabs(root.mn[a]));
        sc.sceneAbs = gmax(sc.sceneAbs, gabs(root.mx[a]));
    }

    std::vector<uint32_t> tmp(N);
    uint32_t              levelStart = 0, levelEnd = 1, depth = 1;
    while(levelStart < levelEnd)
    {
        uint32_t next = levelEnd;
        for(uint32_t ni = levelStart; ni < levelEnd; ++ni)
        {
            const uint32_t start = sc.nodes[ni].first, count = sc.nodes[ni].count;
            if(count <= LEAF_MAX)
                continue;  // leaf
            float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
            for(uint32_t i = start; i < start + count; ++i)
                for(int a = 0; a < 3; ++a)
                {
                    const float c = cen[(size_t)sc.order[i] * 3 + a];
                    cmin[a]       = kmin(cmin[a], c);
                    cmax[a]       = kmax(cmax[a], c);
                }
            float    bestCost = INFINITY;
            int      bestAxis = -1, bestSplit = 0;
            float    bestScale = 0.0f;
            uint32_t bestNL    = 0;
            Box      bestL, bestR;
            for(int a = 0; a < 3; ++a)
            {
                const float ext = cmax[a] - cmin[a];
                if(!(ext > 0.0f))
                    continue;
                const float scale = (float)NB / ext;
                uint32_t    cnt[NB];
                Box         bb[NB];
                for(int b = 0; b < NB; ++b)
                {
                    cnt[b] = 0;
                    boxEmpty(bb[b]);
                }
                for(uint32_t i = start; i < start + count; ++i)
                {
                    const uint32_t p = sc.order[i];
                    int            b = (int)((cen[(size_t)p * 3 + a] - cmin[a]) * scale);
                    if(b > NB - 1)
                        b = NB - 1;
                    cnt[b]++;
                    boxGrow(bb[b], pb[p]);
                }
                // suffix boxes
                Box      R[NB];
                uint32_t nRs[NB];
                Box      acc;
                boxEmpty(acc);
                uint32_t nacc = 0;
                for(int b = NB - 1; b >= 1; --b)
                {
                    boxGrow(acc, bb[b]);
                    nacc += cnt[b];
                    R[b]   = acc;
                    nRs[b] = nacc;
                }
                Box L;
                boxEmpty(L);
                uint32_t nL = 0;
                for(int s = 1; s < NB; ++s)
                {
                    boxGrow(L, bb[s - 1]);
                    nL += cnt[s - 1];
                    const uint32_t nR = nRs[s];
                    if(nL == 0 || nR == 0)
                        continue;
                    const float cost = (boxArea(L) * (float)nL) + (boxArea(R[s]) * (float)nR);
                    if(cost < bestCost)
                    {
                        bestCost  = cost;
                        bestAxis  = a;
                        bestSplit = s;
                        bestScale = scale;
                        bestNL    = nL;
                        bestL     = L;
                        bestR     = R[s];
                    }
                }
            }
            uint32_t nL;
            Box      bl, br;
            if(bestAxis < 0)
            {
                // no usable split plane: cut the range in the middle, keep the order
                nL = count / 2;
                boxEmpty(bl);
                boxEmpty(br);
                for(uint32_t i = 0; i < count; ++i)
                    boxGrow(i < nL ? bl : br, pb[sc.order[start + i]]);
            }
            else
            {
                nL = bestNL;
                bl = bestL;
                br = bestR;
                uint32_t l = 0, r = nL;
                for(uint32_t i = start; i < start + count; ++i)
                {
                    const uint32_t p = sc.order[i];
                    int            b = (int)((cen[(size_t)p * 3 + bestAxis] - cmin[bestAxis]) * bestScale);
                    if(b > NB - 1)
                        b = NB - 1;
                    if(b < bestSplit)
                        tmp[l++] = p;
                    else
                        tmp[r++] = p;
                }
                memcpy(&sc.order[start], tmp.data(), (size_t)count * 4);
            }
            orc_node cl, cr;
            memcpy(cl.bmin, bl.mn, 12);
            memcpy(cl.bmax, bl.mx, 12);
            cl.first = start;
            cl.count = nL;
            memcpy(cr.bmin, br.mn, 12);
            memcpy(cr.bmax, br.mx, 12);
            cr.first = start + nL;
            cr.count = count - nL;
            sc.nodes[ni].first = next;
            sc.nodes[ni].count = 0;
            sc.nodes.push_back(cl);
            sc.nodes.push_back(cr);
            next += 2;
        }
        levelStart = levelEnd;
        levelEnd   = next;
        if(levelStart < levelEnd)
            depth++;
    }
    sc.depth = depth;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// traceNV — watertight ray/triangle test (Woop, Benthin, Wald 2013) and a conservative slab test.
// PARITY UNPINNED (the reference's traversal runs inside the NVIDIA driver / RT cores).
// Triangles are two-sided (VK_GEOMETRY_INSTANCE_TRIANGLE_CULL_DISABLE, vkRT_TLAS.cpp:218-223),
// all geometry opaque; a hit needs tmin < t < tmax; closest hit = smallest t, ties -> lower prim id.
// ------------------------------------------------------------------------------------------------
struct RayPre
{
    float o[3], d[3];
    int   kx, ky, kz;
    float Sx, Sy, Sz;
    // slab test
    float onear[3], ofar[3], rdn[3], rdf[3];
    int   neg[3];
    float amax;  // max(|scene|, |origin|): scale of the pads
};

inline void rayPrepare(float sceneAbs, const float o[3], const float d[3], RayPre& r)
{
    for(int k = 0; k < 3; ++k)
    {
        r.o[k] = o[k];
        r.d[k] = d[k];
    }
    int kz = 0;
    if(gabs(d[1]) > gabs(d[kz]))
        kz = 1;
    if(gabs(d[2]) > gabs(d[kz]))
        kz = 2;
    int kx = kz + 1;
    if(kx == 3)
        kx = 0;
    int ky = kx + 1;
    if(ky == 3)
        ky = 0;
    if(d[kz] < 0.0f)
    {
        int t = kx;
        kx    = ky;
        ky    = t;
    }
    r.kx = kx;
    r.ky = ky;
    r.kz = kz;
    r.Sx = d[kx] / d[kz];
    r.Sy = d[ky] / d[kz];
    r.Sz = 1.0f / d[kz];
    // conservative slab set-up: boxes are grown by `pad` and the slab distances widened by 4 ulp
    float amax = sceneAbs;
    for(int k = 0; k < 3; ++k)
        amax = gmax(amax, gabs(o[k]));
    const float pad = 3.814697265625e-06f * amax;  // 64 * 2^-24
    r.amax          = amax;
    for(int k = 0; k < 3; ++k)
    {
        float dk = d[k];
        if(gabs(dk) < 1e-20f)
            dk = (dk < 0.0f) ? -1e-20f : 1e-20f;
        const float rd = 1.0f / dk;
        r.neg[k]       = dk < 0.0f;
        const float sp = r.neg[k] ? -pad : pad;
        r.onear[k]     = o[k] + sp;
        r.ofar[k]      = o[k] - sp;
        r.rdn[k]       = rd * 0.99999976f;
        r.rdf[k]       = rd * 1.00000024f;
    }
}

// returns true if [tn, tf] overlaps; tnOut = entry distance
inline bool slab(const RayPre& r, const float bmin[3], const float bmax[3], float tmin, float tbest, float& tnOut)
{
    float tn = tmin, tf = tbest;
    for(int k = 0; k < 3; ++k)
    {
        const float pn = r.neg[k] ? bmax[k] : bmin[k];
        const float pf = r.neg[k] ? bmin[k] : bmax[k];
        const float a  = (pn - r.onear[k]) * r.rdn[k];
        const float b  = (pf - r.ofar[k]) * r.rdf[k];
        tn             = gmax(tn, a);
        tf             = gmin(tf, b);
    }
    tnOut = tn;
    return tn <= tf;
}

inline float comp(const float* v, int k) { return v[k]; }

// returns true and (t,u,v) when the triangle is hit with tmin < t < tmax
inline bool triTest(const RayPre& r, const float* p0, const float* p1, const float* p2, float tmin, float tmax,
                    float& tOut, float& uOut, float& vOut)
{
    float A[3], B[3], C[3];
    for(int k = 0; k < 3; ++k)
    {
        A[k] = p0[k] - r.o[k];
        B[k] = p1[k] - r.o[k];
        C[k] = p2[k] - r.o[k];
    }
    const float Ax = A[r.kx] - r.Sx * A[r.kz];
    const float Ay = A[r.ky] - r.Sy * A[r.kz];
    const float Bx = B[r.kx] - r.Sx * B[r.kz];
    const float By = B[r.ky] - r.Sy * B[r.kz];
    const float Cx = C[r.kx] - r.Sx * C[r.kz];
    const float Cy = C[r.ky] - r.Sy * C[r.kz];
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
    const float Az  = r.Sz * A[r.kz];
    const float Bz  = r.Sz * B[r.kz];
    const float Cz  = r.Sz * C[r.kz];
    const float T   = (U * Az + V * Bz) + W * Cz;
    const float rcp = 1.0f / det;
    const float t   = T * rcp;
    if(!(t > tmin && t < tmax))
        return false;
    // A ray that grazes the triangle's plane can pass the edge tests with weights U, V, W that are rounding noise, so
    // that o + t*d lies outside the triangle; no conservative box test can be made to keep such a "hit". A hit
    // therefore has to put the hit point, rebuilt from the same weights (depth along the major axis = weighted vertex
    // depths; minor axes = shear * depth), within the triangle's extent along the two minor axes, grown by a quarter
    // of the pad of the slab test (definition shared with the HIP kernel).
    {
        const float hpad = 9.5367431640625e-07f * r.amax;  // 16 * 2^-24
        const float zh   = ((U * A[r.kz] + V * B[r.kz]) + W * C[r.kz]) * rcp;
        const float xh = r.Sx * zh, yh = r.Sy * zh;
        if(xh < gmin(gmin(A[r.kx], B[r.kx]), C[r.kx]) - hpad || xh > gmax(gmax(A[r.kx], B[r.kx]), C[r.kx]) + hpad)
            return false;
        if(yh < gmin(gmin(A[r.ky], B[r.ky]), C[r.ky]) - hpad || yh > gmax(gmax(A[r.ky], B[r.ky]), C[r.ky]) + hpad)
            return false;
    }
    tOut = t;
    uOut = V * rcp;
    vOut = W * rcp;
    return true;
}

inline const float* triVertex(const orc_scene& sc, const Geo& g, uint32_t prim, int k)
{
    return sc.verts[sc.idx[3 * (size_t)(g.firstTri + prim) + k]].p;
}

// Closest hit inside ONE structure, ray given in the structure's space. `best` carries the closest hit so far (and
// bounds the search); primitive ids are reported as primBase + local id.
void traceClosestGeo(const orc_scene& sc, const Geo& g, const float o[3], float tmin, const float d[3], float tmax,
                     bool brute, uint32_t primBase, orc_hit& best)
{
    RayPre r;
    rayPrepare(g.sceneAbs, o, d, r);
    auto testPrim = [&](uint32_t p) {
        float t, u, v;
        if(triTest(r, triVertex(sc, g, p, 0), triVertex(sc, g, p, 1), triVertex(sc, g, p, 2), tmin, tmax, t, u, v))
        {
            if(t < best.t || (t == best.t && primBase + p < best.prim))
            {
                best.t    = t;
                best.u    = u;
                best.v    = v;
                best.prim = primBase + p;
            }
        }
    };
    if(brute)
    {
        for(uint32_t p = 0; p < g.nTri; ++p)
            testPrim(p);
        return;
    }
    uint32_t stack[128];
    int      sp  = 0;
    uint32_t cur = 0;
    for(;;)
    {
        const orc_node& n = g.nodes[cur];
        if(n.count > 0)
        {
            for(uint32_t i = 0; i < n.count; ++i)
                testPrim(g.order[n.first + i]);
        }
        else
        {
            const orc_node& L = g.nodes[n.first];
            const orc_node& R = g.nodes[n.first + 1];
            float           tl, tr;
            const bool      hl = slab(r, L.bmin, L.bmax, tmin, best.t, tl);
            const bool      hr = slab(r, R.bmin, R.bmax, tmin, best.t, tr);
            if(hl && hr)
            {
                if(tr < tl)
                {
                    stack[sp++] = n.first;
                    cur         = n.first + 1;
                }
                else
                {
                    stack[sp++] = n.first + 1;
                    cur         = n.first;
                }
                continue;
            }
            if(hl)
            {
                cur = n.first;
                continue;
            }
            if(hr)
            {
                cur = n.first + 1;
                continue;
            }
        }
        if(sp == 0)
            break;
        cur = stack[--sp];
    }
}

bool traceAnyGeo(const orc_scene& sc, const Geo& g, const float o[3], float tmin, const float d[3], float tmax,
                 bool brute)
{
    RayPre r;
    rayPrepare(g.sceneAbs, o, d, r);
    float t, u, v;
    if(brute)
    {
        for(uint32_t p = 0; p < g.nTri; ++p)
            if(triTest(r, triVertex(sc, g, p, 0), triVertex(sc, g, p, 1), triVertex(sc, g, p, 2), tmin, tmax, t, u, v))
                return true;
        return false;
    }
    uint32_t stack[128];
    int      sp  = 0;
    uint32_t cur = 0;
    for(;;)
    {
        const orc_node& n = g.nodes[cur];
        if(n.count > 0)
        {
            for(uint32_t i = 0; i < n.count; ++i)
            {
                const uint32_t p = g.order[n.first + i];
                if(triTest(r, triVertex(sc, g, p, 0), triVertex(sc, g, p, 1), triVertex(sc, g, p, 2), tmin, tmax, t, u, v))
                    return true;
            }
        }
        else
        {
            const orc_node& L = g.nodes[n.first];
            const orc_node& R = g.nodes[n.first + 1];
            float           tl, tr;
            const bool      hl = slab(r, L.bmin, L.bmax, tmin, tmax, tl);
            const bool      hr = slab(r, R.bmin, R.bmax, tmin, tmax, tr);
            if(hl && hr)
            {
                if(tr < tl)
                {
                    stack[sp++] = n.first;
                    cur         = n.first + 1;
                }
                else
                {
                    stack[sp++] = n.first + 1;
                    cur         = n.first;
                }
                continue;
            }
            if(hl)
            {
                cur = n.first;
                continue;
            }
            if(hr)
            {
                cur = n.first + 1;
                continue;
            }
        }
        if(sp == 0)
            break;
        cur = stack[--sp];
    }
    return false;
}

// ---- instances (two-level scenes) ---------------------------------------------------------------
// An instance is hit where its bottom-level structure is hit by the ray taken to OBJECT space:
//   o' = (W0*o.x + W1*o.y) + (W2*o.z + W3),   d' = (W0*d.x + W1*d.y) + W2*d.z      (W = inverse transform, columns)
// with the same t (d is not normalised, so the parameter is shared between the spaces, as in VK_NV_ray_tracing).
// An instance whose transform is exactly the identity takes the ray as it is. The hit of the scene is the smallest t
// over all instances, ties -> lower flattened primitive id (instance's base + local id). PARITY UNPINNED, like the
// single-level trace: the reference never builds more than one identity instance (vkRTX_setup.cpp:138-186).
inline void rayToObject(const Inst& in, const float o[3], const float d[3], float oo[3], float od[3])
{
    if(in.identity)
    {
        for(int k = 0; k < 3; ++k)
        {
            oo[k] = o[k];
            od[k] = d[k];
        }
        return;
    }
    const float* W = in.w2o;
    for(int k = 0; k < 3; ++k)
    {
        oo[k] = (W[0 + k] * o[0] + W[4 + k] * o[1]) + (W[8 + k] * o[2] + W[12 + k]);
        od[k] = (W[0 + k] * d[0] + W[4 + k] * d[1]) + W[8 + k] * d[2];
    }
}

// conservative "may the world ray touch this instance's box" (plain slab test with a generous margin)
inline bool mayTouch(const Box& b, const float o[3], const float d[3], float tmin, float tmax)
{
    float tn = tmin, tf = tmax;
    for(int k = 0; k < 3; ++k)
    {
        if(o[k] != o[k] || d[k] != d[k])
            return true;  // let the exact test sort out NaN rays
        if(gabs(d[k]) < 1e-20f)
        {
            if(o[k] < b.mn[k] || o[k] > b.mx[k])
                return false;
            continue;
        }
        const double a = ((double)b.mn[k] - (double)o[k]) / (double)d[k], c = ((double)b.mx[k] - (double)o[k]) / (double)d[k];
        const double lo = a < c ? a : c, hi = a < c ? c : a;
        const double m  = 1e-5 * ((lo < 0 ? -lo : lo) + (hi < 0 ? -hi : hi)) + 1e-30;
        if(lo - m > tn)
            tn = (float)(lo - m);
        if(hi + m < tf)
            tf = (float)(hi + m);
    }
    return tn <= tf * 1.00001f + 1e-30f;
}

orc_hit traceClosest(const orc_scene& sc, const float o[3], float tmin, const float d[3], float tmax, bool brute)
{
    orc_hit best;
    best.prim = 0xffffffffu;
    best.u = best.v = 0.0f;
    best.t          = tmax;
    if(sc.inst.empty())
    {
        traceClosestGeo(sc, sc.geo, o, tmin, d, tmax, brute, 0u, best);
        return best;
    }
    for(const Inst& in : sc.inst)
    {
        if(!brute && !mayTouch(in.world, o, d, tmin, best.t))
            continue;
        float oo[3], od[3];
        rayToObject(in, o, d, oo, od);
        traceClosestGeo(sc, sc.blas[in.blas], oo, tmin, od, tmax, brute, in.primBase, best);
    }
    return best;
}

bool traceAny(const orc_scene& sc, const float o[3], float tmin, const float d[3], float tmax, bool brute)
{
    if(sc.inst.empty())
        return traceAnyGeo(sc, sc.geo, o, tmin, d, tmax, brute);
    for(const Inst& in : sc.inst)
    {
        if(!brute && !mayTouch(in.world, o, d, tmin, tmax))
            continue;
        float oo[3], od[3];
        rayToObject(in, o, d, oo, od);
        if(traceAnyGeo(sc, sc.blas[in.blas], oo, tmin, od, tmax, brute))
            return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// shader helpers
// ------------------------------------------------------------------------------------------------
struct Vertex  // pathRT.rgen:94-119
{
    vec3 pos, normal;
    vec2 texCoord;
    int  matIndex;
};
inline Vertex unpackVertex(const orc_scene& sc, uint32_t index)
{
    const prt_vertex& s = sc.verts[index];
    Vertex            v;
    v.pos      = V3(s.p[0], s.p[1], s.p[2]);
    v.normal   = V3(s.n[0], s.n[1], s.n[2]);
    v.texCoord = vec2{s.t[0], s.t[1]};
    v.matIndex = s.materialID;
    return v;
}

// Corner k of the (flattened) primitive `prim` as the shaders see it. Single-level: the scene's vertex. Two-level: the
// primitive is `prim - primBase` of its instance's bottom-level structure, taken to world space the way an instanced
// scene is flattened (pathtracer_amd/host/Scene.cpp): p = (M0*x + M1*y) + (M2*z + M3), n = normalize(N * n) with
// N = transpose(inverse(M)) (a zero normal stays zero); texture coordinates and material unchanged; an identity
// instance is taken bit for bit.
inline const Inst& instanceOf(const orc_scene& sc, uint32_t prim)
{
    size_t lo = 0, hi = sc.inst.size();  // last instance whose primBase <= prim
    while(hi - lo > 1)
    {
        const size_t mid = (lo + hi) / 2;
        if(sc.inst[mid].primBase <= prim)
            lo = mid;
        else
            hi = mid;
    }
    return sc.inst[lo];
}
inline Vertex shadingVertex(const orc_scene& sc, uint32_t prim, int k)
{
    if(sc.inst.empty())
        return unpackVertex(sc, sc.idx[3 * (size_t)prim + k]);
    const Inst& in    = instanceOf(sc, prim);
    const Geo&  g     = sc.blas[in.blas];
    Vertex      v     = unpackVertex(sc, sc.idx[3 * (size_t)(g.firstTri + (prim - in.primBase)) + k]);
    if(!in.identity)
    {
        const float *M = in.o2w, *N = in.nrm;
        const vec3   p = v.pos, n = v.normal;
        v.pos = V3((M[0] * p.x + M[4] * p.y) + (M[8] * p.z + M[12]), (M[1] * p.x + M[5] * p.y) + (M[9] * p.z + M[13]),
                   (M[2] * p.x + M[6] * p.y) + (M[10] * p.z + M[14]));
        vec3 nn = V3((N[0] * n.x + N[4] * n.y) + N[8] * n.z, (N[1] * n.x + N[5] * n.y) + N[9] * n.z,
                     (N[2] * n.x + N[6] * n.y) + N[10] * n.z);
        if(dot(nn, nn) > 0.0f)
            nn = normalize(nn);
        v.normal = nn;
    }
    return v;
}

// pathRT.rgen:176-197, AO.rgen:104-125
inline mat3 formBasis(vec3 n)
{
    vec3 T, B;
    if(n.z < -0.9999999f)
    {
        T = V3(0.0f, -1.0f, 0.0f);
        B = V3(-1.0f, 0.0f, 0.0f);
    }
    else
    {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        T             = V3(1.0f - n.x * n.x * a, b, -n.x);
        B             = V3(b, 1.0f - n.y * n.y * a, -n.y);
    }
    return mat3{T, B, n};
}

// pathRT.rgen:541-567 (AO.rgen:185-200 inlines the same arithmetic with var = 0.5)
inline void getPrimaryRay(const prt_uniforms& ubo, uint32_t px, uint32_t py, uint32_t w, uint32_t h, vec2 var,
                          vec3& Ro, vec3& Rd)
{
    const vec2 pixelCenter = vec2{(float)px + var.x, (float)py + var.y};
    const vec2 inUV        = vec2{pixelCenter.x / (float)w, pixelCenter.y / (float)h};
    const vec2 d           = vec2{inUV.x * 2.0f - 1.0f, inUV.y * 2.0f - 1.0f};
    const vec4 p0          = vec4{d.x, d.y, 0.0f, 1.0f};
    const vec4 p1          = vec4{d.x, d.y, 1.0f, 1.0f};
    const vec4 Roh         = mul4(ubo.projViewInverse, p0);
    const float iw0        = 1.0f / Roh.w;
    Ro                     = V3(Roh.x * iw0, Roh.y * iw0, Roh.z * iw0);
    const vec4 Rdh         = mul4(ubo.projViewInverse, p1);
    const float iw1        = 1.0f / Rdh.w;
    Rd                     = V3(Rdh.x * iw1, Rdh.y * iw1, Rdh.z * iw1);
    Rd                     = Rd - Ro;
}

// texture(sampler2D, uv): bilinear, REPEAT, RGBA8 UNORM, one mip (vkTools.cpp:405,456-480).
// Filter-weight precision is implementation defined in Vulkan -> full fp32 weights here.
inline vec3 textureRGB(const Tex& t, vec2 uv)
{
    const float fx = uv.x * (float)t.w - 0.5f;
    const float fy = uv.y * (float)t.h - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    auto wrap = [](float f, uint32_t n) -> uint32_t {
        // f is integral; bring into [0,n)
        float m = f - floorf(f / (float)n) * (float)n;
        if(!(m == m)) return 0u;  // NaN coordinate: texel 0
        int i = (int)m;
        if(i < 0)
            i += (int)n;
        if(i >= (int)n)
            i -= (int)n;
        if(i < 0) i = 0;  // |f| beyond 2^24 * n
        if(i >= (int)n) i = (int)n - 1;
        return (uint32_t)i;
    };
    const uint32_t x0 = wrap(x0f, t.w), x1 = wrap(x0f + 1.0f, t.w);
    const uint32_t y0 = wrap(y0f, t.h), y1 = wrap(y0f + 1.0f, t.h);
    auto texel = [&](uint32_t x, uint32_t y) -> vec3 {
        const uint8_t* p = &t.rgba[((size_t)y * t.w + x) * 4];
        return V3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
    };
    const vec3 c00 = texel(x0, y0), c10 = texel(x1, y0), c01 = texel(x0, y1), c11 = texel(x1, y1);
    const vec3 top = c00 * (1.0f - ax) + c10 * ax;
    const vec3 bot = c01 * (1.0f - ax) + c11 * ax;
    return top * (1.0f - ay) + bot * ay;
}

// pathRT.rgen:304-309
inline void sampleLight(const prt_uniforms& ubo, float& pdf, vec3& p, vec2 prng)
{
    pdf            = 1.0f / (4.0f * ubo.lightSize[0] * ubo.lightSize[1]);
    const vec2 pos = vec2{(prng.x * 2.0f - 1.0f) * ubo.lightSize[0], (prng.y * 2.0f - 1.0f) * ubo.lightSize[1]};
    const vec4 r   = mul4(ubo.lightTransform, vec4{pos.x, pos.y, 0.0f, 1.0f});
    p              = V3(r.x, r.y, r.z);
}

inline float GGX_chi(float v) { return v > 0.0f ? 1.0f : 0.0f; }  // :326-329

inline float GGX_Distribution(vec3 wm, float alpha)  // :331-341
{
    const float a2    = alpha * alpha;
    const float MdotN = wm.z;
    const float chi   = GGX_chi(MdotN);
    const float cos2  = MdotN * MdotN;
    const float tan2  = (1.0f - cos2) / cos2;
    const float m4    = (MdotN * MdotN) * (MdotN * MdotN);
    const float q     = a2 + tan2;
    const float denom = kPi * m4 * (q * q);
    return chi * a2 / denom;
}

inline float GGX_G1(vec3 wv, vec3 wm, float alpha)  // :343-356
{
    const float a2     = alpha * alpha;
    const float VdotM  = dot(wv, wm);
    const float VdotN  = wv.z;
    const float VdotM2 = VdotM * VdotM;
    const float tan2   = (1.0f - VdotM2) / VdotM2;
    const float chi    = GGX_chi(VdotM / VdotN);
    const float denom  = 1.0f + sqrtf(1.0f + a2 * tan2);
    return chi * 2.0f / denom;
}

inline float GGX_FresnelDielectric(float cosThetaI, float ni, float nt)  // :435-463
{
    cosThetaI = gclamp(cosThetaI, -1.0f, 1.0f);
    if(cosThetaI < 0.0f)
    {
        const float temp = ni;
        ni               = nt;
        nt               = temp;
        cosThetaI        = -cosThetaI;
    }
    const float sinThetaI = sqrtf(gmax(0.0f, 1.0f - cosThetaI * cosThetaI));
    const float sinThetaT = ni / nt * sinThetaI;
    if(sinThetaI >= 1.0f)
        return 1.0f;
    const float cosThetaT      = sqrtf(gmax(0.0f, 1.0f - sinThetaT * sinThetaT));
    const float rParallel      = ((nt * cosThetaI) - (ni * cosThetaT)) / ((nt * cosThetaI) + (ni * cosThetaT));
    const float rPerpendicular = ((ni * cosThetaI) - (nt * cosThetaT)) / ((ni * cosThetaI) + (nt * cosThetaT));
    return (rParallel * rParallel + rPerpendicular * rPerpendicular) * 0.5f;
}

inline vec3 GGX_SchlickFresnel(vec3 r0, float radians)  // :465-469
{
    const float x    = 1.0f - radians;
    const float x2   = x * x;
    const float expo = (x2 * x2) * x;
    return r0 + (V3(1.0f) - r0) * expo;
}

inline float GGX_SmithMasking(vec3 /*wi*/, vec3 wo, float alpha)  // :471-478
{
    const float a2    = alpha * alpha;
    const float NdotV = gabs(wo.z);
    const float denom = sqrtf(a2 + (1.0f - a2) * NdotV * NdotV) + NdotV;
    return 2.0f * NdotV / denom;
}

inline float GGX_SmithMaskingShadowing(vec3 wi, vec3 wo, float alpha)  // :480-490
{
    const float a2     = alpha * alpha;
    const float NdotL  = wi.z;
    const float NdotV  = wo.z;
    const float denomA = NdotV * sqrtf(a2 + (1.0f - a2) * NdotL * NdotL);
    const float denomB = NdotL * sqrtf(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * NdotL * NdotV / gclamp(denomA + denomB, 1e-4f, 1.0f);
}

inline vec3 GGX_SampleVNDF(vec3 wo, float alpha, vec2 rnd)  // :492-509
{
    const vec3 v  = normalize(V3(wo.x * alpha, wo.y * alpha, wo.z));
    const vec3 t1 = (v.z < 0.9999f) ? normalize(cross(v, V3(0.0f, 0.0f, 1.0f))) : V3(1.0f, 0.0f, 0.0f);
    const vec3 t2 = cross(t1, v);
    const float a   = 1.0f / (1.0f + v.z);
    const float r   = sqrtf(rnd.x);
    const float phi = (rnd.y < a) ? rnd.y / a * kPi : kPi + (rnd.y - a) / (1.0f - a) * kPi;
    float       sn, cs;
    sincos_spec(phi, &sn, &cs);
    const float p1 = r * cs;
    const float p2 = r * sn * ((rnd.y < a) ? 1.0f : v.z);
    vec3        n  = (p1 * t1 + p2 * t2) + sqrtf(gmax(0.0f, 1.0f - p1 * p1 - p2 * p2)) * v;
    n              = normalize(V3(alpha * n.x, alpha * n.y, gmax(0.0f, n.z)));
    return n;
}

inline bool refract_(vec3 wi, vec3 wn, vec3& wt, float eta)  // :511-526
{
    const float cosThetaI  = dot(wn, wi);
    const float sin2ThetaI = gmax(0.0f, 1.0f - cosThetaI * cosThetaI);
    const float sin2ThetaT = eta * eta * sin2ThetaI;
    if(sin2ThetaT >= 1.0f)
        return false;
    const float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    wt                    = eta * (-wi) + (eta * cosThetaI - cosThetaT) * wn;
    return true;
}

inline float mitchellNetrevali(float v)  // :575-593
{
    const float B = 1.0f / 3.0f;
    const float C = 1.0f / 3.0f;
    if(v < 1.0f)
    {
        return ((12.0f - 9.0f * B - 6.0f * C) * ((v * v) * v) + (-18.0f + 12.0f * B + 6.0f * C) * (v * v)
                + (6.0f - 2.0f * B))
               / 6.0f;
    }
    return ((-B - 6.0f * C) * ((v * v) * v) + (6.0f * B + 30.0f * C) * (v * v) + (-12.0f * B - 48.0f * C) * v
            + (8.0f * B + 24.0f * C))
           / 6.0f;
}
inline float getMitchellWeight(vec2 offset)  // :595-598
{
    return mitchellNetrevali(offset.x) * mitchellNetrevali(offset.y);
}

// AO.rgen:153-181
inline vec3 hemisphereSample(const orc_scene& sc, uint32_t index, const uint32_t scramble[2])
{
    float s[2];
    for(int d = 0; d < 2; ++d)
        s[d] = sobol1DSample(sc, index, (uint32_t)d, scramble[d]);
    float       phi, r;
    const float a = 2.0f * s[0] - 1.0f;
    const float b = 2.0f * s[1] - 1.0f;
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
    sincos_spec(phi, &sn, &cs);
    const float x = r * cs;
    const float y = r * sn;
    const float z = sqrtf(gmax(0.0f, 1.0f - x * x - y * y));
    return V3(x, y, z);
}

struct PixelCtx
{
    const orc_scene&    sc;
    const prt_uniforms& ubo;
    uint32_t            w, h, flags;
    bool                brute;
    uint64_t            rays = 0;
};

inline void a3(const vec3& v, float o[3])
{
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
}

// ------------------------------------------------------------------------------------------------
// AO.rgen main — AO.rgen:183-273
// ------------------------------------------------------------------------------------------------
void aoPixel(PixelCtx& cx, uint32_t px, uint32_t py, float* Eio, uint32_t* hitId)
{
    const orc_scene&    sc  = cx.sc;
    const prt_uniforms& ubo = cx.ubo;
    vec3                Ro, Rd;
    getPrimaryRay(ubo, px, py, cx.w, cx.h, vec2{0.5f, 0.5f}, Ro, Rd);
    const float tmin = 0.00001f, tmax = 1.0f;
    const bool  literal = (cx.flags & ORC_FLAG_LITERAL) != 0;
    vec4        E{0.0f, 0.0f, 0.0f, 0.0f};
    if(!literal || ubo.iteration > 1)
        E = vec4{Eio[0], Eio[1], Eio[2], Eio[3]};
    float o[3], d[3];
    a3(Ro, o);
    a3(Rd, d);
    const orc_hit hit = traceClosest(sc, o, tmin, d, tmax, cx.brute);
    cx.rays++;
    if(hitId)
        *hitId = hit.prim;
    if(hit.prim == 0xffffffffu)
    {
        if(literal)
        {
            Eio[0] = 0.1f; Eio[1] = 0.1f; Eio[2] = 0.1f; Eio[3] = 0.0f;
        }
        else
        {
            Eio[0] = E.x + 0.1f; Eio[1] = E.y + 0.1f; Eio[2] = E.z + 0.1f; Eio[3] = E.w + 0.0f;
        }
        return;
    }
    uint32_t scramble[2];
    scramble[0]         = scrambleValue(sc, px, py, cx.w, cx.h, 0);
    scramble[1]         = scrambleValue(sc, px, py, cx.w, cx.h, 1);
    uint32_t sobolIndex = scrambleValue(sc, px, py, cx.w, cx.h, 2);

    const Vertex v0 = shadingVertex(sc, hit.prim, 0);
    const Vertex v1 = shadingVertex(sc, hit.prim, 1);
    const Vertex v2 = shadingVertex(sc, hit.prim, 2);
    const vec3   bary = V3(1.0f - hit.u - hit.v, hit.u, hit.v);
    vec3         normal = normalize((v0.normal * bary.x + v1.normal * bary.y) + v2.normal * bary.z);
    if(dot(normal, Rd) > 0.0f)
        normal = normal * -1.0f;
    const mat3 ONB      = formBasis(normal);
    vec3       hitPoint = (v0.pos * bary.x + v1.pos * bary.y) + v2.pos * bary.z;
    hitPoint            = hitPoint + 0.0001f * normal;

    int aoNoHitCount = 0;
    a3(hitPoint, o);
    for(int i = 0; i < ubo.numAOrays; ++i)
    {
        const vec3 v = hemisphereSample(sc, sobolIndex++, scramble);
        Rd           = normalize(mul(ONB, v)) * V3(ubo.aoRayLength);
        a3(Rd, d);
        const bool isShadowed = traceAny(sc, o, tmin, d, tmax, cx.brute);
        cx.rays++;
        if(!isShadowed)
            aoNoHitCount++;
    }
    const float ao = (float)aoNoHitCount / (float)ubo.numAOrays;
    Eio[0]         = E.x + ao;
    Eio[1]         = E.y + ao;
    Eio[2]         = E.z + ao;
    Eio[3]         = E.w + 1.0f;
}

// ------------------------------------------------------------------------------------------------
// pathRT.rgen main — pathRT.rgen:621-974
// ------------------------------------------------------------------------------------------------
// ORC_DEBUG_PIXEL="x,y": print every ray pathPixel traces for that pixel and what it hit (test infrastructure only)
static bool debugPixel(uint32_t px, uint32_t py)
{
    static int dx = -2, dy = -2;
    if(dx == -2)
    {
        dx = dy = -1;
        if(const char* e = getenv("ORC_DEBUG_PIXEL")) sscanf(e, "%d,%d", &dx, &dy);
    }
    return (int)px == dx && (int)py == dy;
}
static void debugRay(const char* what, int aa, int bounce, const float* o, float tmin, const float* d, float tmax,
                     uint32_t prim, float t)
{
    fprintf(stderr, "ORC %s aa %d bounce %d o %a %a %a tmin %a d %a %a %a tmax %a -> prim %u t %a\n", what, aa, bounce,
            o[0], o[1], o[2], tmin, d[0], d[1], d[2], tmax, prim, t);
}

void pathPixel(PixelCtx& cx, uint32_t px, uint32_t py, float* Eio, uint32_t* hitId)
{
    const bool dbg = debugPixel(px, py);
    const orc_scene&    sc  = cx.sc;
    const prt_uniforms& ubo = cx.ubo;
    const bool          literal = (cx.flags & ORC_FLAG_LITERAL) != 0;
    const bool          rr      = (cx.flags & ORC_FLAG_RUSSIAN_ROULETTE) != 0;

    const vec2 pixelCenter = vec2{(float)px + 0.5f, (float)py + 0.5f};
    const vec2 inUV        = vec2{pixelCenter.x / (float)cx.w, pixelCenter.y / (float)cx.h};

    uint32_t sobolIndex = scrambleValue(sc, px, py, cx.w, cx.h, 0);
    sobolIndex /= 2;
    sobolIndex += ubo.iteration;
    uint32_t sobolDim = 0;

    uint32_t scrambleArray[16][2];
    for(int i = 0; i < 16; i++)
    {
        scrambleArray[i][0] = scrambleValue(sc, px, py, cx.w, cx.h, 2 * i + 0);
        scrambleArray[i][1] = scrambleValue(sc, px, py, cx.w, cx.h, 2 * i + 1);
    }

    const float tmin       = 0.000001f;
    const float tmax       = 1.0f;
    const int   maxBounces = ubo.numIndirectBounces;
    vec4        E{0.0f, 0.0f, 0.0f, 0.0f};
    if(!literal || ubo.iteration > 1)
        E = vec4{Eio[0], Eio[1], Eio[2], Eio[3]};

    float o[3], d[3];
    for(int aaRay = 0; aaRay < ubo.numAArays; ++aaRay)
    {
        vec3     Ro, Rd;
        vec2     rayOffset          = vec2{0.5f, 0.5f};
        vec3     throughput         = V3(1.0f);
        float    p                  = 1.0f;
        int      bounce             = 0;
        uint32_t scrambleArrayLayer = 1;
        sobolDim                    = 2;

        rayOffset = nextSquareSample(sc, sobolIndex, sobolDim, scrambleArray[0]);
        if(ubo.numAArays == 1)
        {
            rayOffset = vec2{rayOffset.x * 2.0f - 1.0f, rayOffset.y * 2.0f - 1.0f};
            rayOffset = rayOffset * ubo.filterRadius;
            getPrimaryRay(ubo, px, py, cx.w, cx.h, rayOffset + vec2{0.5f, 0.5f}, Ro, Rd);
        }
        else
        {
            getPrimaryRay(ubo, px, py, cx.w, cx.h, rayOffset, Ro, Rd);
        }

        a3(Ro, o);
        a3(Rd, d);
        orc_hit payload = traceClosest(sc, o, tmin, d, tmax, cx.brute);
        cx.rays++;
        if(dbg) debugRay("primary", aaRay, 0, o, tmin, d, tmax, payload.prim, payload.t);
        if(hitId)
            *hitId = payload.prim;
        if(payload.prim == 0xffffffffu)
        {
            if(literal)
            {
                Eio[0] = inUV.x; Eio[1] = inUV.y; Eio[2] = 0.4f; Eio[3] = 1.0f;
            }
            else
            {
                Eio[0] = E.x + inUV.x; Eio[1] = E.y + inUV.y; Eio[2] = E.z + 0.4f; Eio[3] = E.w + 1.0f;
            }
            return;
        }

        while(true)
        {
            const uint32_t primitiveID  = payload.prim;
            const vec3     barycentrics = V3(1.0f - payload.u - payload.v, payload.u, payload.v);  // pathRT.rchit:19
            const uint32_t* scramble    = scrambleArray[scrambleArrayLayer++];

            const Vertex v0 = shadingVertex(sc, primitiveID, 0);
            const Vertex v1 = shadingVertex(sc, primitiveID, 1);
            const Vertex v2 = shadingVertex(sc, primitiveID, 2);

            // out-of-range material / texture ids are undefined behaviour in the shader; defined here as
            // "material 0" / "no texture".
            const prt_material& mat =
                sc.mats[(v1.matIndex >= 0 && (size_t)v1.matIndex < sc.mats.size()) ? v1.matIndex : 0];
            const int ntex = (int)sc.tex.size();

            vec3 sNormal =
                normalize((v0.normal * barycentrics.x + v1.normal * barycentrics.y) + v2.normal * barycentrics.z);
            if(dot(sNormal, Rd) > 0.0f)
                sNormal = -sNormal;
            const vec3 gNormal = sNormal;

            const vec3 hitPoint = (v0.pos * barycentrics.x + v1.pos * barycentrics.y) + v2.pos * barycentrics.z;

            vec3 albedo   = V3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
            vec3 specular = V3(mat.specular[0], mat.specular[1], mat.specular[2]);
            vec3 textureN = V3(1.0f);

            const vec2 texCoord = vec2{(v0.texCoord.x * barycentrics.x + v1.texCoord.x * barycentrics.y)
                                           + v2.texCoord.x * barycentrics.z,
                                       (v0.texCoord.y * barycentrics.x + v1.texCoord.y * barycentrics.y)
                                           + v2.texCoord.y * barycentrics.z};
            if(mat.diffuseTextureID >= 0 && mat.diffuseTextureID < ntex)
                albedo = albedo * textureRGB(sc.tex[mat.diffuseTextureID], texCoord);
            if(mat.specularTextureID >= 0 && mat.specularTextureID < ntex)
                specular = specular * textureRGB(sc.tex[mat.specularTextureID], texCoord);
            if(mat.normalTextureID >= 0 && mat.normalTextureID < ntex)
                textureN = textureRGB(sc.tex[mat.normalTextureID], texCoord) * 2.0f - V3(1.0f);

            // bump mapping — :754-783
            if(mat.normalTextureID >= 0 && mat.normalTextureID < ntex)
            {
                const vec3  v10  = v1.pos - v0.pos;
                const vec3  v20  = v2.pos - v0.pos;
                const vec2  uv10 = v1.texCoord - v0.texCoord;
                const vec2  uv20 = v2.texCoord - v0.texCoord;
                // M = [uv10 | uv20] (columns); determinant = m00*m11 - m10*m01
                const float det = uv10.x * uv20.y - uv20.x * uv10.y;
                if(gabs(det) > 1e-4f)
                {
                    const float oneOverDet = 1.0f / det;
                    // inverse(mat2): [ m11, -m01 ; -m10, m00 ] * 1/det, column major
                    const vec2 col0 = vec2{uv20.y * oneOverDet, -uv10.y * oneOverDet};
                    const vec2 col1 = vec2{-uv20.x * oneOverDet, uv10.x * oneOverDet};
                    const vec3 T    = normalize(v10 * col0.x + v20 * col0.y);
                    const vec3 B    = normalize(v10 * col1.x + v20 * col1.y);
                    const mat3 normalM{T, B, sNormal};
                    sNormal = mul(normalM, textureN);
                }
            }

            {  // :786-793
                const float mx = maxcoord(albedo + specular);
                if(mx > 1.0f)
                {
                    albedo   = albedo / V3(mx);
                    specular = specular / V3(mx);
                }
            }

            float pdf;
            vec3  lightSamplePos;
            {
                const vec2 s = nextSquareSample(sc, sobolIndex, sobolDim, scramble);
                sampleLight(ubo, pdf, lightSamplePos, s);
            }
            const vec3 vLight = lightSamplePos - hitPoint;

            const float roughness = gclamp(1.0f - mat.shininess, 0.001f, 1.0f);
            const float alpha     = roughness * roughness;

            const mat3 mLocalToWorld = formBasis(gNormal);
            const vec3 wo            = normalize(mulT(mLocalToWorld, -Rd));

            vec3 Ei = V3(0.0f);

            a3(hitPoint, o);
            a3(vLight, d);
            const bool isShadowed = traceAny(sc, o, tmin, d, tmax, cx.brute);
            cx.rays++;
            if(dbg) debugRay("shadow", aaRay, bounce, o, tmin, d, tmax, isShadowed ? 1u : 0xffffffffu, 0.0f);
            if(!isShadowed)
            {
                const vec4  l2 = vec4{ubo.lightTransform[8], ubo.lightTransform[9], ubo.lightTransform[10], 0.0f};
                const vec3  lightNormal      = -normalize(V3(l2.x, l2.y, l2.z));
                const float r                = length(vLight);
                const float cosTheta_light   = gclamp(dot(normalize(-vLight), lightNormal), 0.0f, 1.0f);
                const float cosTheta_surface = gclamp(dot(sNormal, normalize(vLight)), 0.0f, 1.0f);
                const vec3  wi               = normalize(mulT(mLocalToWorld, vLight));
                const vec3  wm               = normalize(wo + vLight);  // sic: local + world (pathRT.rgen:837)
                const vec3  F                = GGX_SchlickFresnel(specular, dot(wi, wm));
                const float G1               = GGX_SmithMasking(wi, wo, alpha);
                const float G2               = GGX_SmithMaskingShadowing(wi, wo, alpha);
                const vec3  brdf             = F * (G2 / G1);
                const vec3  lightE           = V3(ubo.lightE[0], ubo.lightE[1], ubo.lightE[2]);
                Ei = Ei + lightE * cosTheta_light * cosTheta_surface / (r * r * pdf);
                Ei = Ei * (brdf * throughput);
            }

            vec3 wi           = V3(0.0f);
            vec3 wm           = V3(0.0f);
            vec3 specularBRDF = V3(0.0f);
            {
                const vec2 rnd = nextSquareSample(sc, sobolIndex, sobolDim, scramble);
                wm             = GGX_SampleVNDF(wo, alpha, rnd);

                const float fres = GGX_FresnelDielectric(wo.z, 1.0f, 1.45f);
                if(mat.dissolve == 0.0f)
                {
                    if(rnd.x < fres)
                    {
                        wi = normalize(reflect(-wo, wm));
                    }
                    else
                    {
                        if(refract_(wo, V3(0.0f, 0.0f, 1.0f), wi, 1.0f / 1.45f))
                        {
                        }
                        else
                        {
                            wi = reflect(-wo, wm);
                        }
                        wi = normalize(wi);
                    }
                }
                else
                {
                    wi = normalize(reflect(-wo, wm));
                }

                if(wi.z > 0.0f)
                {
                    if(mat.dissolve == 1.0f)
                    {
                        const vec3  F   = GGX_SchlickFresnel(specular, dot(wi, wm));
                        const float G1  = gclamp(GGX_SmithMasking(wi, wo, alpha), 1e-4f, 1.0f);
                        const float G2  = GGX_SmithMaskingShadowing(wi, wo, alpha);
                        const float D   = GGX_Distribution(wm, alpha);
                        const float G1v = GGX_G1(wo, wm, alpha);
                        specular        = F * G1v;
                        specularBRDF    = F * (G2 / G1);
                        p               = G1v * gabs(dot(wm, wi)) * D / wi.z;
                        p *= 1.0f / (4.0f * gabs(dot(wo, wm)));
                    }
                    else
                    {
                        const float D   = GGX_Distribution(wm, alpha);
                        const float G1v = GGX_G1(wo, wm, alpha);
                        pdf             = 1.0f - gclamp(fres, 0.0f, 1.0f - 1e-2f);
                        specularBRDF    = albedo / pdf;
                        p               = G1v * gabs(dot(wm, wi)) * D / wi.z;
                        p *= 1.0f / (4.0f * gabs(dot(wo, wm)));
                    }
                }
            }

            wi = mul(mLocalToWorld, wi);
            Ro = hitPoint;
            Rd = wi * V3(1000.0f);

            const vec3 diffuse = albedo * 1.0f;
            vec4       color;
            {
                const vec3 c = diffuse + specular * specularBRDF;
                color        = vec4{c.x, c.y, c.z, 1.0f};
            }
            if(ubo.numAArays != 1)
            {
                const float weight = getMitchellWeight(rayOffset + vec2{0.5f, 0.5f});
                color              = vec4{color.x / weight, color.y / weight, color.z / weight, color.w / weight};
            }

            Ei = Ei + V3(mat.emission[0], mat.emission[1], mat.emission[2]) * ubo.lightOtherE;
            Ei = Ei * (throughput * V3(color.x, color.y, color.z));
            E  = vec4{E.x + Ei.x, E.y + Ei.y, E.z + Ei.z, E.w + color.w};

            if(!(p < 0.0f || p >= 0.0f))
                break;
            if(p == 0.0f)
                break;
            throughput = throughput * V3(color.x, color.y, color.z);

            bounce++;
            if(bounce > maxBounces)
                break;

            if(rr && bounce >= 2)
            {
                // Russian roulette — extension (north_star); NOT in the reference, off by default.
                const float q  = gclamp(maxcoord(throughput), 0.05f, 1.0f);
                const float xi = sobol1DSample(sc, sobolIndex, 48u + (uint32_t)bounce, scrambleArray[15][bounce & 1]);
                if(xi >= q)
                    break;
                throughput = throughput / V3(q);
            }

            a3(Ro, o);
            a3(Rd, d);
            payload = traceClosest(sc, o, tmin, d, tmax, cx.brute);
            cx.rays++;
            if(dbg) debugRay("bounce", aaRay, bounce, o, tmin, d, tmax, payload.prim, payload.t);
            if(payload.prim == 0xffffffffu)
                break;
        }
        sobolIndex++;
    }
    Eio[0] = E.x;
    Eio[1] = E.y;
    Eio[2] = E.z;
    Eio[3] = E.w;
}

template <class F>
void parallelRows(uint32_t n, int threads, F&& f)
{
    if(threads <= 0)
        threads = (int)std::thread::hardware_concurrency();
    if(threads < 1)
        threads = 1;
    std::atomic<uint32_t>    next{0};
    std::vector<std::thread> pool;
    auto                     work = [&](int tid) {
        for(;;)
        {
            const uint32_t i = next.fetch_add(1);
            if(i >= n)
                break;
            f(i, tid);
        }
    };
    for(int t = 1; t < threads; ++t)
        pool.emplace_back(work, t);
    work(0);
    for(auto& t : pool)
        t.join();
}

}  // namespace

// max |coordinate| over the triangles of a range: what buildBVH derives from the root box, available before a build
// so that brute-force traces use the same pads
static float rangeAbs(const orc_scene& sc, uint32_t first, uint32_t n)
{
    float a = 0.0f;
    for(size_t i = (size_t)3 * first; i < (size_t)3 * (first + n); ++i)
        for(int k = 0; k < 3; ++k)
            a = gmax(a, gabs(sc.verts[sc.idx[i]].p[k]));
    return a;
}

// ------------------------------------------------------------------------------------------------
// C interface
// ------------------------------------------------------------------------------------------------
extern "C" {

orc_scene* orc_scene_create(const prt_vertex* v, size_t nv, const uint32_t* idx, size_t nidx,
                            const prt_material* m, size_t nm, const prt_texture* tex, size_t ntex)
{
    orc_scene* sc = new orc_scene();
    sc->verts.assign(v, v + nv);
    sc->idx.assign(idx, idx + (nidx / 3) * 3);
    sc->mats.assign(m, m + nm);
    for(size_t i = 0; i < ntex; ++i)
    {
        Tex t;
        t.w = tex[i].width;
        t.h = tex[i].height;
        t.rgba.assign(tex[i].rgba, tex[i].rgba + (size_t)t.w * t.h * 4);
        sc->tex.push_back(std::move(t));
    }
    sc->geo.firstTri = 0;
    sc->geo.nTri     = (uint32_t)sc->ntri();
    sc->geo.sceneAbs = rangeAbs(*sc, 0, sc->geo.nTri);
    return sc;
}
void orc_scene_destroy(orc_scene* sc) { delete sc; }

void orc_set_sobol(orc_scene* sc, const uint32_t* matrices, uint32_t dims, uint32_t size)
{
    sc->sobol.assign(matrices, matrices + (size_t)dims * size);
    sc->sobolDims = dims;
    sc->sobolSize = size;
}
void orc_set_scramble_seed(orc_scene* sc, uint32_t seed)
{
    sc->seed     = seed;
    sc->scrTable = nullptr;
}
void orc_set_scramble_table(orc_scene* sc, const uint32_t* table, uint32_t w, uint32_t h)
{
    sc->scrTable = table;
    sc->scrW     = w;
    sc->scrH     = h;
}

// glm 0.9.9.5 detail/func_matrix.inl compute_inverse<4,4> (cofactors, then * 1/determinant), same operation order;
// a = column-major 4x4. The host layer's prt_math.h restates the same function and is pinned against glm itself
// (tests/golden/host_golden.json); tests/test_oracle_cpu.py compares the two.
void orc_inverse4(const float* a, float* out)
{
#define M(c, r) a[(c)*4 + (r)]
    const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    const float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3), c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    const float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2), c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    const float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2), c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, V1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    const float V2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, V3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    float       inv[16];
    for(int j = 0; j < 4; ++j)
    {
        const float sa = (j & 1) ? -1.0f : 1.0f, sb = -sa;
        inv[0 + j]  = ((V1[j] * F0[j] - V2[j] * F1[j]) + V3[j] * F2[j]) * sa;
        inv[4 + j]  = ((V0[j] * F0[j] - V2[j] * F3[j]) + V3[j] * F4[j]) * sb;
        inv[8 + j]  = ((V0[j] * F1[j] - V1[j] * F3[j]) + V3[j] * F5[j]) * sa;
        inv[12 + j] = ((V0[j] * F2[j] - V1[j] * F4[j]) + V2[j] * F5[j]) * sb;
    }
    const float d0 = M(0, 0) * inv[0], d1 = M(0, 1) * inv[4], d2 = M(0, 2) * inv[8], d3 = M(0, 3) * inv[12];
    const float det = (d0 + d1) + (d2 + d3);
    const float ood = 1.0f / det;
    for(int k = 0; k < 16; ++k)
        out[k] = inv[k] * ood;
#undef M
}

/* Two-level scene: `blas` = ranges of the scene's triangles (object space), `inst` = instances of them. Primitive
 * ids reported by traces and renders are flattened: instance i's triangles follow those of instances 0..i-1.
 * n_inst == 0 returns to the single-level scene. Call orc_build_bvh afterwards. Returns 0, or -1 on a bad range /
 * index. */
int orc_set_instances(orc_scene* sc, const prt_blas_range* blas, size_t n_blas, const prt_instance* inst, size_t n_inst)
{
    sc->blas.clear();
    sc->inst.clear();
    if(n_inst == 0)
        return 0;
    for(size_t k = 0; k < n_blas; ++k)
    {
        if(blas[k].n_triangles == 0 || (size_t)blas[k].first_triangle + blas[k].n_triangles > sc->ntri())
            return -1;
        Geo g;
        g.firstTri = blas[k].first_triangle;
        g.nTri     = blas[k].n_triangles;
        g.sceneAbs = rangeAbs(*sc, g.firstTri, g.nTri);
        sc->blas.push_back(std::move(g));
    }
    uint64_t base = 0;
    for(size_t i = 0; i < n_inst; ++i)
    {
        if(inst[i].blas >= n_blas)
        {
            sc->blas.clear();
            sc->inst.clear();
            return -1;
        }
        Inst in;
        memcpy(in.o2w, inst[i].transform, 64);
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        in.identity = memcmp(in.o2w, ident, 64) == 0;
        orc_inverse4(in.o2w, in.w2o);
        for(int c = 0; c < 4; ++c)
            for(int r = 0; r < 4; ++r)
                in.nrm[c * 4 + r] = in.w2o[r * 4 + c];
        in.blas       = inst[i].blas;
        in.instanceId = inst[i].instance_id;
        in.primBase   = (uint32_t)base;
        base += sc->blas[in.blas].nTri;
        boxEmpty(in.world);
        sc->inst.push_back(in);
    }
    return base < 0xffffffffull ? 0 : -1;
}

int orc_build_bvh(orc_scene* sc)
{
    if(sc->inst.empty())
    {
        sc->geo.firstTri = 0;
        sc->geo.nTri     = (uint32_t)sc->ntri();
        return buildBVH(*sc, sc->geo);
    }
    for(Geo& g : sc->blas)
        if(buildBVH(*sc, g))
            return -1;
    // world boxes of the instances: the 8 corners of the structure's box, transformed in double, with a margin
    for(Inst& in : sc->inst)
    {
        const orc_node& root = sc->blas[in.blas].nodes[0];
        boxEmpty(in.world);
        double ext = 0.0;
        for(int c = 0; c < 8; ++c)
        {
            const double p[3] = {(c & 1) ? root.bmax[0] : root.bmin[0], (c & 2) ? root.bmax[1] : root.bmin[1],
                                 (c & 4) ? root.bmax[2] : root.bmin[2]};
            for(int k = 0; k < 3; ++k)
            {
                const double w = (double)in.o2w[0 + k] * p[0] + (double)in.o2w[4 + k] * p[1] + (double)in.o2w[8 + k] * p[2] +
                                 (double)in.o2w[12 + k];
                if(w < in.world.mn[k]) in.world.mn[k] = (float)w;
                if(w > in.world.mx[k]) in.world.mx[k] = (float)w;
                ext = std::fmax(ext, std::fabs(w));
            }
        }
        const float m = (float)(ext * 1e-3 + 1e-6);
        for(int k = 0; k < 3; ++k)
        {
            in.world.mn[k] = std::nextafter(in.world.mn[k], -INFINITY) - m;
            in.world.mx[k] = std::nextafter(in.world.mx[k], INFINITY) + m;
        }
    }
    return 0;
}
uint32_t orc_bvh_num_nodes(const orc_scene* sc) { return (uint32_t)sc->geo.nodes.size(); }
uint32_t orc_bvh_depth(const orc_scene* sc) { return sc->geo.depth; }
void     orc_bvh_read(const orc_scene* sc, orc_node* nodes_out, uint32_t* order_out)
{
    if(nodes_out)
        memcpy(nodes_out, sc->geo.nodes.data(), sc->geo.nodes.size() * sizeof(orc_node));
    if(order_out)
        memcpy(order_out, sc->geo.order.data(), sc->geo.order.size() * 4);
}
/* the same for bottom-level structure k of a two-level scene (order: ids local to the structure's range) */
uint32_t orc_blas_num_nodes(const orc_scene* sc, uint32_t k) { return k < sc->blas.size() ? (uint32_t)sc->blas[k].nodes.size() : 0u; }
void     orc_blas_read(const orc_scene* sc, uint32_t k, orc_node* nodes_out, uint32_t* order_out)
{
    if(k >= sc->blas.size())
        return;
    if(nodes_out)
        memcpy(nodes_out, sc->blas[k].nodes.data(), sc->blas[k].nodes.size() * sizeof(orc_node));
    if(order_out)
        memcpy(order_out, sc->blas[k].order.data(), sc->blas[k].order.size() * 4);
}

float orc_sobol_sample(const orc_scene* sc, uint32_t index, uint32_t dim, uint32_t scramble)
{
    return sobol1DSample(*sc, index, dim, scramble);
}
uint32_t orc_scramble(const orc_scene* sc, uint32_t px, uint32_t py, uint32_t w, uint32_t h, uint32_t layer)
{
    return scrambleValue(*sc, px, py, w, h, layer);
}
void orc_sincos(float x, float* s, float* c) { sincos_spec(x, s, c); }

void orc_primary_ray(const prt_uniforms* ubo, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float varx,
                     float vary, orc_ray* out)
{
    vec3 Ro, Rd;
    getPrimaryRay(*ubo, px, py, w, h, vec2{varx, vary}, Ro, Rd);
    a3(Ro, out->o);
    a3(Rd, out->d);
}

void orc_trace_closest(const orc_scene* sc, const orc_ray* rays, size_t n, orc_hit* out, uint32_t flags,
                       int threads)
{
    const bool     brute = (flags & ORC_FLAG_BRUTE_FORCE) != 0;
    const uint32_t chunk = 256;
    const uint32_t nch   = (uint32_t)((n + chunk - 1) / chunk);
    parallelRows(nch, threads, [&](uint32_t c, int) {
        const size_t e = (size_t)(c + 1) * chunk < n ? (size_t)(c + 1) * chunk : n;
        for(size_t i = (size_t)c * chunk; i < e; ++i)
            out[i] = traceClosest(*sc, rays[i].o, rays[i].tmin, rays[i].d, rays[i].tmax, brute);
    });
}

void orc_trace_any(const orc_scene* sc, const orc_ray* rays, size_t n, uint8_t* occluded, uint32_t flags,
                   int threads)
{
    const bool     brute = (flags & ORC_FLAG_BRUTE_FORCE) != 0;
    const uint32_t chunk = 256;
    const uint32_t nch   = (uint32_t)((n + chunk - 1) / chunk);
    parallelRows(nch, threads, [&](uint32_t c, int) {
        const size_t e = (size_t)(c + 1) * chunk < n ? (size_t)(c + 1) * chunk : n;
        for(size_t i = (size_t)c * chunk; i < e; ++i)
            occluded[i] = traceAny(*sc, rays[i].o, rays[i].tmin, rays[i].d, rays[i].tmax, brute) ? 1 : 0;
    });
}

uint64_t orc_render(const orc_scene* sc, const prt_uniforms* ubo, uint32_t w, uint32_t h, int mode, uint32_t flags,
                    float* accum, uint32_t* hit_ids, int threads)
{
    std::atomic<uint64_t> total{0};
    parallelRows(h, threads, [&](uint32_t y, int) {
        PixelCtx cx{*sc, *ubo, w, h, flags, (flags & ORC_FLAG_BRUTE_FORCE) != 0};
        for(uint32_t x = 0; x < w; ++x)
        {
            const size_t i  = (size_t)y * w + x;
            uint32_t*    id = hit_ids ? &hit_ids[i] : nullptr;
            if(mode == 1)
                aoPixel(cx, x, y, &accum[i * 4], id);
            else
                pathPixel(cx, x, y, &accum[i * 4], id);
        }
        total.fetch_add(cx.rays);
    });
    return total.load();
}

// pow(x, 1/2.2) of the post-process (pathRTpostprocess.comp:48). GLSL leaves the rounding of pow open; this is the
// definition the build uses, the same expression tree in the oracle and in the kernel, so the RGBA8 image is bit-exact:
// x = m 2^e with m in [sqrt(1/2), sqrt(2)), log2 m by the atanh series in s = (m-1)/(m+1) up to s^9, y = log2(x) / 2.2,
// 2^y = 2^n exp(f ln 2) with the Taylor polynomial of degree 7 on |f| <= 1/2 (relative error < 3e-7).
// x <= 0 or NaN gives 0 (the UNORM store would clamp them to 0 anyway).
static inline int float_bits(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float bits_float(int i) { float f; memcpy(&f, &i, 4); return f; }
float orc_gamma(float x)
{
    if(!(x > 0.0f)) return 0.0f;
    if(x > 3.0e38f) return x;
    float scale_e = 0.0f;
    if(x < 1.17549435e-38f)
    {
        x       = x * 16777216.0f;
        scale_e = -24.0f;
    }
    const int ix = float_bits(x);
    int       e  = ((ix >> 23) & 0xff) - 127;
    float     m  = bits_float((ix & 0x007fffff) | 0x3f800000);
    if(m > 1.41421356f)
    {
        m = m * 0.5f;
        e = e + 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    const float p = ((((z * 0.111111111f + 0.142857143f) * z + 0.2f) * z + 0.333333333f) * z) + 1.0f;
    const float l = ((float)e + scale_e) + (2.88539008f * s) * p;
    const float y = l * 0.454545455f;
    const float n = floorf(y + 0.5f);
    const float t = (y - n) * 0.693147181f;
    const float q = ((((((t * 1.98412698e-4f + 1.38888889e-3f) * t + 8.33333333e-3f) * t + 4.16666667e-2f) * t + 0.166666667f) * t + 0.5f) * t + 1.0f) * t + 1.0f;
    return bits_float(float_bits(q) + (int)n * (1 << 23));  // (not "<< 23": n may be negative)
}

// The shading helpers one by one, for the identity tests (unit tier) and for the transcription audit
// against the shader text compiled with glm (oracle/make_glsl_audit.py). `which`: ORC_H_* of oracle.h; `in` / `out`:
// the function's arguments / results as flat float arrays, in the order of the GLSL signature.
int orc_helper(int which, const float* in, float* out)
{
    auto v3 = [&](int k) { return V3(in[k], in[k + 1], in[k + 2]); };
    auto put3 = [&](int k, vec3 v) { out[k] = v.x; out[k + 1] = v.y; out[k + 2] = v.z; };
    switch(which)
    {
    case ORC_H_FORM_BASIS:
    {
        const mat3 m = formBasis(v3(0));
        put3(0, m.c0); put3(3, m.c1); put3(6, m.c2);
        return 9;
    }
    case ORC_H_SAMPLE_LIGHT:  // in: lightSize[2], lightTransform[16], prng[2]
    {
        prt_uniforms u;
        memset(&u, 0, sizeof u);
        u.lightSize[0] = in[0]; u.lightSize[1] = in[1];
        memcpy(u.lightTransform, in + 2, 64);
        float pdf = 0.0f;
        vec3  p   = V3(0.0f);
        sampleLight(u, pdf, p, vec2{in[18], in[19]});
        out[0] = pdf; put3(1, p);
        return 4;
    }
    case ORC_H_GGX_DISTRIBUTION: out[0] = GGX_Distribution(v3(0), in[3]); return 1;
    case ORC_H_GGX_G1: out[0] = GGX_G1(v3(0), v3(3), in[6]); return 1;
    case ORC_H_FRESNEL_DIELECTRIC: out[0] = GGX_FresnelDielectric(in[0], in[1], in[2]); return 1;
    case ORC_H_SCHLICK_FRESNEL: put3(0, GGX_SchlickFresnel(v3(0), in[3])); return 3;
    case ORC_H_SMITH_MASKING: out[0] = GGX_SmithMasking(v3(0), v3(3), in[6]); return 1;
    case ORC_H_SMITH_MASKING_SHADOWING: out[0] = GGX_SmithMaskingShadowing(v3(0), v3(3), in[6]); return 1;
    case ORC_H_SAMPLE_VNDF: put3(0, GGX_SampleVNDF(v3(0), in[3], vec2{in[4], in[5]})); return 3;
    case ORC_H_REFRACT:
    {
        vec3 wt = V3(0.0f);
        out[0]  = refract_(v3(0), v3(3), wt, in[6]) ? 1.0f : 0.0f;
        put3(1, wt);
        return 4;
    }
    case ORC_H_PRIMARY_RAY:  // in: projViewInverse[16], px, py, w, h, var[2]
    {
        prt_uniforms u;
        memset(&u, 0, sizeof u);
        memcpy(u.projViewInverse, in, 64);
        vec3 Ro, Rd;
        getPrimaryRay(u, (uint32_t)in[16], (uint32_t)in[17], (uint32_t)in[18], (uint32_t)in[19], vec2{in[20], in[21]}, Ro, Rd);
        put3(0, Ro); put3(3, Rd);
        return 6;
    }
    case ORC_H_MITCHELL: out[0] = mitchellNetrevali(in[0]); return 1;
    case ORC_H_MITCHELL_WEIGHT: out[0] = getMitchellWeight(vec2{in[0], in[1]}); return 1;
    default: return -1;
    }
}

// texture(textureSamplers[id], uv).xyz as the oracle defines it (textureRGB above): for the transcription audit
void orc_texture_rgb(const orc_scene* s, int id, float u, float v, float* rgb_out)
{
    vec3 c = V3(1.0f);
    if(s && id >= 0 && (size_t)id < s->tex.size()) c = textureRGB(s->tex[(size_t)id], vec2{u, v});
    rgb_out[0] = c.x;
    rgb_out[1] = c.y;
    rgb_out[2] = c.z;
}

// pathRTpostprocess.comp:38-51
void orc_postprocess(const float* accum, uint32_t w, uint32_t h, uint8_t* out)
{
    for(size_t i = 0; i < (size_t)w * h; ++i)
    {
        float c[4] = {accum[4 * i], accum[4 * i + 1], accum[4 * i + 2], accum[4 * i + 3]};
        if(c[3] != 0.0f)
            for(int k = 0; k < 4; ++k)
                c[k] /= c[3];
        for(int k = 0; k < 3; ++k)
            c[k] = orc_gamma(c[k]);
        for(int k = 0; k < 4; ++k)
        {
            float v = c[k];
            v       = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);  // UNORM store clamps; NaN -> 0
            out[4 * i + k] = (uint8_t)(v * 255.0f + 0.5f);
        }
    }
}

}  // extern "C"
