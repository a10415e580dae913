// prt_device.hpp — device-side math of the MI355X ray-tracing path (gfx950, wave64).
//
// GLSL built-ins and the reference's shader helper functions, written out as explicit fp32
// expression trees. The translation unit is compiled with -ffp-contract=off: every operator is
// one IEEE-754 rounding, in source order, so results do not depend on instruction selection.
//
// Evaluation-order conventions (DESIGN.md "Numerics"):
//   dot(a,b)     = (a.x*b.x + a.y*b.y) + a.z*b.z
//   mat3*vec3    = (c0*x + c1*y) + c2*z           mat4*vec4 = (c0*x + c1*y) + (c2*z + c3*w)
//   normalize(v) = v * (1/sqrt(dot(v,v)))
//   min/max      = GLSL: min(x,y) = y<x ? y : x,  max(x,y) = x<y ? y : x
//   pow(x,n)     = products for the constant exponents the shaders use
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/prt_types.h"

#define PRT_DEV __device__ __forceinline__

namespace prt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct m3 { f3 c0, c1, c2; };

PRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PRT_DEV f3 mk3(float s) { return mk3(s, s, s); }
PRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PRT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

PRT_DEV float gmin(float x, float y) { return (y < x) ? y : x; }
PRT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }
PRT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
PRT_DEV float gabs(float x) { return __builtin_fabsf(x); }
PRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PRT_DEV f3    cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
PRT_DEV float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
PRT_DEV f3    normalize3(f3 a)
{
    const float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * inv;
}
PRT_DEV f3    reflect3(f3 I, f3 N) { return I - (2.0f * dot3(N, I)) * N; }
PRT_DEV f3    mulM(const m3& m, f3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
PRT_DEV f3    mulMT(const m3& m, f3 v) { return mk3(dot3(m.c0, v), dot3(m.c1, v), dot3(m.c2, v)); }
PRT_DEV float maxcoord(f3 v) { return gmax(gmax(v.x, v.y), v.z); }
PRT_DEV void  mul44(const float* __restrict__ m, float x, float y, float z, float w, float out[4])
{
    out[0] = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
    out[1] = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
    out[2] = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
    out[3] = (m[3] * x + m[7] * y) + (m[11] * z + m[15] * w);
}

// Streaming (read-once / write-once) per-ray records: non-temporal, so that they do not push the BVH out of L2.
typedef float    v4f_t __attribute__((ext_vector_type(4)));
typedef uint32_t v1u_t;
PRT_DEV float4 ld_nt(const float4* p)
{
    const v4f_t v = __builtin_nontemporal_load((const v4f_t*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
PRT_DEV void st_nt(float4* p, float4 q)
{
#ifdef PRT_ST_PLAIN
    *p = q;
#else
    v4f_t v;
    v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    __builtin_nontemporal_store(v, (v4f_t*)p);
#endif
}
// Results of a traversal (hit records, radiance terms) are stored with the default policy: a finished lane's store
// sits in the same in-order vmcnt queue as the node loads that follow, and a write-back store is acknowledged by the
// L2 sooner than a streaming one (measured: +1.5 % on the bench; the same switch in k_shade costs 7 % there, because
// its stores are many and would push the shading triangles out of L2). PRT_TRACE_ST_NT: measurement switch.
PRT_DEV void st_res(float4* p, float4 q)
{
#ifdef PRT_TRACE_ST_NT
    st_nt(p, q);
#else
    *p = q;
#endif
}
PRT_DEV uint32_t ld_nt(const uint32_t* p) { return __builtin_nontemporal_load(p); }
PRT_DEV void     st_nt(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

constexpr float kPi = 3.141592653589f;  // pathRT.rgen:85

// sin/cos with a fixed polynomial: Cody-Waite reduction by pi/2 + single-precision minimax kernels.
// GLSL does not define the rounding of sin/cos; this is the definition the render uses.
PRT_DEV void sincos_fixed(float x, float& s_out, float& c_out)
{
    const float q = __builtin_floorf(x * 0.636619772f + 0.5f);
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
    const int   m = k & 3;
    s_out         = (m == 0) ? s : (m == 1) ? c : (m == 2) ? -s : -c;
    c_out         = (m == 0) ? c : (m == 1) ? -s : (m == 2) ? -c : s;
}

// ---- random numbers ------------------------------------------------------------------------
// Sobol' matrices for 64 dimensions x 32 index bits (table built on the host from the Joe-Kuo
// direction numbers, sobol_table.hpp). `dim` is wave-uniform in every kernel, so the matrix words
// are fetched through the scalar cache.   sobol1DSample: shaders/pathRT.rgen:220-235
PRT_DEV float sobol1D(const uint32_t* __restrict__ M, uint32_t index, uint32_t dim, uint32_t scramble)
{
    uint32_t        result = scramble;
    const uint32_t* row    = M + dim * 32u;
#pragma unroll
    for(int i = 0; i < 32; ++i)
    {
        const uint32_t w = row[i];
        result ^= ((index >> i) & 1u) ? w : 0u;
    }
    return (float)result * 2.3283064365386963e-10f;
}

// The same sample from byte tables: T[b][v] = XOR of the matrix words 8b+j over the set bits j of v, so
// four table reads replace the 32-step loop (XOR is exact: bit-identical). `T` is the dimension's
// [4][256] table, staged in LDS by the shade kernel.
PRT_DEV float sobol1D_bytes(const uint32_t* T, uint32_t index, uint32_t scramble)
{
    const uint32_t result = scramble ^ T[index & 255u] ^ T[256u + ((index >> 8) & 255u)]
                            ^ T[512u + ((index >> 16) & 255u)] ^ T[768u + (index >> 24)];
    return (float)result * 2.3283064365386963e-10f;
}

PRT_DEV uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// Stateless replacement for the 32 mt19937 scramble layers (vkRTX_setup.cpp:808-832): no HBM traffic
// (the reference fetches 33 texels = 132 B per pixel per iteration, pathRT.rgen:627-642).
PRT_DEV uint32_t scramble_value(const uint32_t* __restrict__ table, uint32_t npix, uint32_t seed, uint32_t pixel,
                                uint32_t layer)
{
    if(table)
        return table[(size_t)layer * npix + pixel];
    const uint32_t x = lowbias32(pixel * 32u + layer + 0x9E3779B9u * seed);
    return lowbias32(x ^ (seed + 0x85EBCA6Bu));
}

// ---- shading helpers -------------------------------------------------------------------------
// formBasis — pathRT.rgen:176-197
PRT_DEV m3 form_basis(f3 n)
{
    f3 T, B;
    if(n.z < -0.9999999f)
    {
        T = mk3(0.0f, -1.0f, 0.0f);
        B = mk3(-1.0f, 0.0f, 0.0f);
    }
    else
    {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        T             = mk3(1.0f - n.x * n.x * a, b, -n.x);
        B             = mk3(b, 1.0f - n.y * n.y * a, -n.y);
    }
    m3 r;
    r.c0 = T;
    r.c1 = B;
    r.c2 = n;
    return r;
}

// getPrimaryRay — pathRT.rgen:541-567
PRT_DEV void primary_ray(const float* __restrict__ invP, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float varx,
                         float vary, f3& Ro, f3& Rd)
{
    const float cx = (float)px + varx, cy = (float)py + vary;
    const float ux = cx / (float)w, uy = cy / (float)h;
    const float dx = ux * 2.0f - 1.0f, dy = uy * 2.0f - 1.0f;
    float       a[4], b[4];
    mul44(invP, dx, dy, 0.0f, 1.0f, a);
    const float iw0 = 1.0f / a[3];
    Ro              = mk3(a[0] * iw0, a[1] * iw0, a[2] * iw0);
    mul44(invP, dx, dy, 1.0f, 1.0f, b);
    const float iw1 = 1.0f / b[3];
    Rd              = mk3(b[0] * iw1, b[1] * iw1, b[2] * iw1);
    Rd              = Rd - Ro;
}

PRT_DEV float ggx_chi(float v) { return v > 0.0f ? 1.0f : 0.0f; }  // :326-329

PRT_DEV float ggx_distribution(f3 wm, float alpha)  // :331-341
{
    const float a2    = alpha * alpha;
    const float MdotN = wm.z;
    const float chi   = ggx_chi(MdotN);
    const float cos2  = MdotN * MdotN;
    const float tan2  = (1.0f - cos2) / cos2;
    const float m4    = (MdotN * MdotN) * (MdotN * MdotN);
    const float q     = a2 + tan2;
    const float denom = kPi * m4 * (q * q);
    return chi * a2 / denom;
}

PRT_DEV float ggx_g1(f3 wv, f3 wm, float alpha)  // :343-356
{
    const float a2     = alpha * alpha;
    const float VdotM  = dot3(wv, wm);
    const float VdotN  = wv.z;
    const float VdotM2 = VdotM * VdotM;
    const float tan2   = (1.0f - VdotM2) / VdotM2;
    const float chi    = ggx_chi(VdotM / VdotN);
    const float denom  = 1.0f + __builtin_sqrtf(1.0f + a2 * tan2);
    return chi * 2.0f / denom;
}

PRT_DEV float ggx_fresnel_dielectric(float cosThetaI, float ni, float nt)  // :435-463
{
    cosThetaI = gclamp(cosThetaI, -1.0f, 1.0f);
    if(cosThetaI < 0.0f)
    {
        const float temp = ni;
        ni               = nt;
        nt               = temp;
        cosThetaI        = -cosThetaI;
    }
    const float sinThetaI = __builtin_sqrtf(gmax(0.0f, 1.0f - cosThetaI * cosThetaI));
    const float sinThetaT = ni / nt * sinThetaI;
    if(sinThetaI >= 1.0f)
        return 1.0f;
    const float cosThetaT      = __builtin_sqrtf(gmax(0.0f, 1.0f - sinThetaT * sinThetaT));
    const float rParallel      = ((nt * cosThetaI) - (ni * cosThetaT)) / ((nt * cosThetaI) + (ni * cosThetaT));
    const float rPerpendicular = ((ni * cosThetaI) - (nt * cosThetaT)) / ((ni * cosThetaI) + (nt * cosThetaT));
    return (rParallel * rParallel + rPerpendicular * rPerpendicular) * 0.5f;
}

PRT_DEV f3 ggx_schlick(f3 r0, float radians)  // :465-469
{
    const float x    = 1.0f - radians;
    const float x2   = x * x;
    const float expo = (x2 * x2) * x;
    return r0 + (mk3(1.0f) - r0) * expo;
}

PRT_DEV float ggx_smith_masking(f3 wo, float alpha)  // :471-478 (wi is unused there)
{
    const float a2    = alpha * alpha;
    const float NdotV = gabs(wo.z);
    const float denom = __builtin_sqrtf(a2 + (1.0f - a2) * NdotV * NdotV) + NdotV;
    return 2.0f * NdotV / denom;
}

PRT_DEV float ggx_smith_masking_shadowing(f3 wi, f3 wo, float alpha)  // :480-490
{
    const float a2     = alpha * alpha;
    const float NdotL  = wi.z;
    const float NdotV  = wo.z;
    const float denomA = NdotV * __builtin_sqrtf(a2 + (1.0f - a2) * NdotL * NdotL);
    const float denomB = NdotL * __builtin_sqrtf(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * NdotL * NdotV / gclamp(denomA + denomB, 1e-4f, 1.0f);
}

PRT_DEV f3 ggx_sample_vndf(f3 wo, float alpha, f2 rnd)  // :492-509
{
    const f3    v   = normalize3(mk3(wo.x * alpha, wo.y * alpha, wo.z));
    const f3    t1  = (v.z < 0.9999f) ? normalize3(cross3(v, mk3(0.0f, 0.0f, 1.0f))) : mk3(1.0f, 0.0f, 0.0f);
    const f3    t2  = cross3(t1, v);
    const float a   = 1.0f / (1.0f + v.z);
    const float r   = __builtin_sqrtf(rnd.x);
    const float phi = (rnd.y < a) ? rnd.y / a * kPi : kPi + (rnd.y - a) / (1.0f - a) * kPi;
    float       sn, cs;
    sincos_fixed(phi, sn, cs);
    const float p1 = r * cs;
    const float p2 = r * sn * ((rnd.y < a) ? 1.0f : v.z);
    f3          n  = (p1 * t1 + p2 * t2) + __builtin_sqrtf(gmax(0.0f, 1.0f - p1 * p1 - p2 * p2)) * v;
    n              = normalize3(mk3(alpha * n.x, alpha * n.y, gmax(0.0f, n.z)));
    return n;
}

PRT_DEV bool refract3(f3 wi, f3 wn, f3& wt, float eta)  // :511-526
{
    const float cosThetaI  = dot3(wn, wi);
    const float sin2ThetaI = gmax(0.0f, 1.0f - cosThetaI * cosThetaI);
    const float sin2ThetaT = eta * eta * sin2ThetaI;
    if(sin2ThetaT >= 1.0f)
        return false;
    const float cosThetaT = __builtin_sqrtf(1.0f - sin2ThetaT);
    wt                    = eta * (-wi) + (eta * cosThetaI - cosThetaT) * wn;
    return true;
}

PRT_DEV float mitchell_netrevali(float v)  // :575-593
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

// pow(x, 1/2.2) of the post-process (pathRTpostprocess.comp:48). GLSL leaves the rounding of pow open; this is the
// definition the build uses, the same expression tree in the oracle and in the kernel, so the RGBA8 image is bit-exact:
// x = m 2^e with m in [sqrt(1/2), sqrt(2)), log2 m by the atanh series in s = (m-1)/(m+1) up to s^9, y = log2(x) / 2.2,
// 2^y = 2^n exp(f ln 2) with the Taylor polynomial of degree 7 on |f| <= 1/2 (relative error < 3e-7).
// x <= 0 or NaN gives 0 (the UNORM store would clamp them to 0 anyway).
PRT_DEV float gamma_fixed(float x)
{
    if(!(x > 0.0f)) return 0.0f;
    if(x > 3.0e38f) return x;
    float scale_e = 0.0f;
    if(x < 1.17549435e-38f)
    {
        x       = x * 16777216.0f;
        scale_e = -24.0f;
    }
    const int ix = __float_as_int(x);
    int       e  = ((ix >> 23) & 0xff) - 127;
    float     m  = __int_as_float((ix & 0x007fffff) | 0x3f800000);
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
    const float n = __builtin_floorf(y + 0.5f);
    const float t = (y - n) * 0.693147181f;
    const float q = ((((((t * 1.98412698e-4f + 1.38888889e-3f) * t + 8.33333333e-3f) * t + 4.16666667e-2f) * t + 0.166666667f) * t + 0.5f) * t + 1.0f) * t + 1.0f;
    return __int_as_float(__float_as_int(q) + (int)n * (1 << 23));  // (not "<< 23": n may be negative)
}

// Bilinear / REPEAT fetch of an RGBA8 UNORM texture (sampler of vkTools.cpp:456-480), fp32 weights.
struct TexDesc
{
    const uint8_t* rgba;
    uint32_t       w, h;
};
PRT_DEV uint32_t tex_wrap(float f, uint32_t n)
{
    const float m = f - __builtin_floorf(f / (float)n) * (float)n;
    if(!(m == m)) return 0u;  // NaN coordinate: texel 0 (the float -> int conversion of a NaN is not defined)
    int i = (int)m;
    if(i < 0)
        i += (int)n;
    if(i >= (int)n)
        i -= (int)n;
    // |f| beyond 2^24 * n: one period does not bring the index home; stay inside the texture
    if(i < 0) i = 0;
    if(i >= (int)n) i = (int)n - 1;
    return (uint32_t)i;
}
PRT_DEV f3 tex_texel(const TexDesc& t, uint32_t x, uint32_t y)
{
    const uint32_t v = *(const uint32_t*)(t.rgba + ((size_t)y * t.w + x) * 4);
    return mk3((float)(v & 255u) / 255.0f, (float)((v >> 8) & 255u) / 255.0f, (float)((v >> 16) & 255u) / 255.0f);
}
PRT_DEV f3 texture_rgb(const TexDesc& t, f2 uv)
{
    const float    fx  = uv.x * (float)t.w - 0.5f;
    const float    fy  = uv.y * (float)t.h - 0.5f;
    const float    x0f = __builtin_floorf(fx), y0f = __builtin_floorf(fy);
    const float    ax = fx - x0f, ay = fy - y0f;
    const uint32_t x0 = tex_wrap(x0f, t.w), x1 = tex_wrap(x0f + 1.0f, t.w);
    const uint32_t y0 = tex_wrap(y0f, t.h), y1 = tex_wrap(y0f + 1.0f, t.h);
    const f3       c00 = tex_texel(t, x0, y0), c10 = tex_texel(t, x1, y0);
    const f3       c01 = tex_texel(t, x0, y1), c11 = tex_texel(t, x1, y1);
    const f3       top = c00 * (1.0f - ax) + c10 * ax;
    const f3       bot = c01 * (1.0f - ax) + c11 * ax;
    return top * (1.0f - ay) + bot * ay;
}

}  // namespace prt
