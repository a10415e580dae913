// prt_math.h — the small part of glm 0.9.9.5 that the reference's host code uses on the ray-tracing path
// (CameraControls.cpp:35-41,111-126; vkContext.cpp:1000-1006), as plain fp32 code with the same operation
// order, so that the 464-byte parameter block comes out bit-identical (checked against values produced by
// glm itself: tests/golden/host_golden.json).
//   column-major mat4: m[col][row], as glm and GLSL.
//
// perspectiveRH_ZO, lookAtRH and inverse(mat4) below restate, term for term (that is the point: same roundings), the
// formulas of OpenGL Mathematics (GLM) 0.9.9.5 — ext/matrix_clip_space.inl, ext/matrix_transform.inl,
// detail/func_matrix.inl (compute_inverse<4, 4>). GLM is Copyright (c) 2005 - G-Truc Creation and distributed under
// the MIT License:
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation
//   the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and
//   to permit persons to whom the Software is furnished to do so, subject to the following conditions:
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of
//   the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO
//   THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE
//   AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF
//   CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS
//   IN THE SOFTWARE.
#pragma once
#include <cmath>

namespace prt_host {

struct vec2 { float x = 0, y = 0; };
struct vec3
{
    float x = 0, y = 0, z = 0;
    vec3() {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct vec4
{
    float x = 0, y = 0, z = 0, w = 0;
    vec4() {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float&       operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec4 operator*(const vec4& a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator*(const vec4& a, const vec4& b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4& a, const vec4& b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// glm::radians: degrees * 0.0174532925...
inline float radians(float deg) { return deg * static_cast<float>(0.01745329251994329576923690768489); }
// detail/func_geometric.inl: dot = x + y + z of the products, cross, normalize = v * (1 / sqrt(dot))
inline float dot(const vec3& a, const vec3& b)
{
    const vec3 t(a.x * b.x, a.y * b.y, a.z * b.z);
    return t.x + t.y + t.z;
}
inline vec3 cross(const vec3& x, const vec3& y)
{
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline vec3 normalize(const vec3& v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct mat3
{
    vec3        c[3];
    vec3&       operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};

struct mat4
{
    vec4 c[4];
    mat4() {}
    explicit mat4(float d)
    {
        c[0] = vec4(d, 0, 0, 0);
        c[1] = vec4(0, d, 0, 0);
        c[2] = vec4(0, 0, d, 0);
        c[3] = vec4(0, 0, 0, d);
    }
    vec4&       operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
    const float* data() const { return &c[0].x; }
};

// detail/type_mat4x4.inl operator*(mat4, mat4)
inline mat4 operator*(const mat4& m1, const mat4& m2)
{
    mat4 r;
    for(int j = 0; j < 4; ++j) r[j] = m1[0] * m2[j][0] + m1[1] * m2[j][1] + m1[2] * m2[j][2] + m1[3] * m2[j][3];
    return r;
}
inline mat4 operator*(const mat4& m, float s)
{
    mat4 r;
    for(int j = 0; j < 4; ++j) r[j] = m[j] * s;
    return r;
}

// ext/matrix_clip_space.inl perspectiveRH_ZO (GLM_FORCE_DEPTH_ZERO_TO_ONE, right-handed)
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar)
{
    const float tanHalfFovy = std::tan(fovy / 2.0f);
    mat4        r(0.0f);
    r[0][0] = 1.0f / (aspect * tanHalfFovy);
    r[1][1] = 1.0f / (tanHalfFovy);
    r[2][2] = zFar / (zNear - zFar);
    r[2][3] = -1.0f;
    r[3][2] = -(zFar * zNear) / (zFar - zNear);
    return r;
}

// ext/matrix_transform.inl lookAtRH
inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up)
{
    const vec3 f(normalize(center - eye));
    const vec3 s(normalize(cross(f, up)));
    const vec3 u(cross(s, f));
    mat4       r(1.0f);
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

// detail/func_matrix.inl compute_inverse<4,4>: cofactors, then * (1 / determinant)
inline mat4 inverse(const mat4& m)
{
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const vec4  f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11), f3(c12, c12, c14, c15),
        f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    const vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]),
        v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const vec4 i0(v1 * f0 - v2 * f1 + v3 * f2), i1(v0 * f0 - v2 * f3 + v3 * f4), i2(v0 * f1 - v1 * f3 + v3 * f5),
        i3(v0 * f2 - v1 * f4 + v2 * f5);
    const vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat4       inv;
    inv[0] = i0 * sa;
    inv[1] = i1 * sb;
    inv[2] = i2 * sa;
    inv[3] = i3 * sb;
    const vec4  row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    const vec4  d0(m[0] * row0);
    const float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    return inv * (1.0f / d1);
}
inline mat4 transpose(const mat4& m)
{
    mat4 r;
    for(int i = 0; i < 4; ++i)
        for(int j = 0; j < 4; ++j) r[i][j] = m[j][i];
    return r;
}
// gtc/matrix_inverse.inl inverseTranspose(mat4) — only ever called on the identity model matrix here
inline mat4 inverseTranspose(const mat4& m) { return transpose(inverse(m)); }

}  // namespace prt_host
