// sobol_table.hpp — Sobol' generator matrices for the first 64 dimensions, built at start-up from
// the Joe & Kuo direction numbers (S. Joe, F. Y. Kuo, "Constructing Sobol sequences with better
// two-dimensional projections", SIAM J. Sci. Comput. 30, 2008; data set new-joe-kuo-6.21201).
//
// The reference uploads a precomputed 1024 x 52 table (external/NVIDIA_RTX/sobol/sobol.cpp, uploaded
// at src/vkRTX_setup.cpp:1102-1144) and indexes it as sm[dim * 52 + bit] (shaders/pathRT.rgen:220-235).
// The path tracer never uses more than 60 dimensions nor more than 32 index bits
// (sobolIndex is a uint), so the MI355X path keeps 64 x 32 words = 8 KiB, small enough to stay in the
// scalar cache. tests/test_sobol.py checks the generated words against the reference's table.
#pragma once
#include <stdint.h>

namespace prt {

struct JoeKuoRow
{
    uint8_t  s;      // degree of the primitive polynomial
    uint16_t a;      // its inner coefficients, Joe-Kuo encoding
    uint16_t m[10];  // initial direction numbers m_1..m_s
};

// dimension 0 is the van der Corput sequence (s = 0)
static const JoeKuoRow kJoeKuo[64] = {
    {0, 0, {0}}, // dim 0
    {1, 0, {1}}, // dim 1
    {2, 1, {1, 3}}, // dim 2
    {3, 1, {1, 3, 1}}, // dim 3
    {3, 2, {1, 1, 1}}, // dim 4
    {4, 1, {1, 1, 3, 3}}, // dim 5
    {4, 4, {1, 3, 5, 13}}, // dim 6
    {5, 2, {1, 1, 5, 5, 17}}, // dim 7
    {5, 4, {1, 1, 5, 5, 5}}, // dim 8
    {5, 7, {1, 1, 7, 11, 19}}, // dim 9
    {5, 11, {1, 1, 5, 1, 1}}, // dim 10
    {5, 13, {1, 1, 1, 3, 11}}, // dim 11
    {5, 14, {1, 3, 5, 5, 31}}, // dim 12
    {6, 1, {1, 3, 3, 9, 7, 49}}, // dim 13
    {6, 13, {1, 1, 1, 15, 21, 21}}, // dim 14
    {6, 16, {1, 3, 1, 13, 27, 49}}, // dim 15
    {6, 19, {1, 1, 1, 15, 7, 5}}, // dim 16
    {6, 22, {1, 3, 1, 15, 13, 25}}, // dim 17
    {6, 25, {1, 1, 5, 5, 19, 61}}, // dim 18
    {7, 1, {1, 3, 7, 11, 23, 15, 103}}, // dim 19
    {7, 4, {1, 3, 7, 13, 13, 15, 69}}, // dim 20
    {7, 7, {1, 1, 3, 13, 7, 35, 63}}, // dim 21
    {7, 8, {1, 3, 5, 9, 1, 25, 53}}, // dim 22
    {7, 14, {1, 3, 1, 13, 9, 35, 107}}, // dim 23
    {7, 19, {1, 3, 1, 5, 27, 61, 31}}, // dim 24
    {7, 21, {1, 1, 5, 11, 19, 41, 61}}, // dim 25
    {7, 28, {1, 3, 5, 3, 3, 13, 69}}, // dim 26
    {7, 31, {1, 1, 7, 13, 1, 19, 1}}, // dim 27
    {7, 32, {1, 3, 7, 5, 13, 19, 59}}, // dim 28
    {7, 37, {1, 1, 3, 9, 25, 29, 41}}, // dim 29
    {7, 41, {1, 3, 5, 13, 23, 1, 55}}, // dim 30
    {7, 42, {1, 3, 7, 3, 13, 59, 17}}, // dim 31
    {7, 50, {1, 3, 1, 3, 5, 53, 69}}, // dim 32
    {7, 55, {1, 1, 5, 5, 23, 33, 13}}, // dim 33
    {7, 56, {1, 1, 7, 7, 1, 61, 123}}, // dim 34
    {7, 59, {1, 1, 7, 9, 13, 61, 49}}, // dim 35
    {7, 62, {1, 3, 3, 5, 3, 55, 33}}, // dim 36
    {8, 14, {1, 3, 1, 15, 31, 13, 49, 245}}, // dim 37
    {8, 21, {1, 3, 5, 15, 31, 59, 63, 97}}, // dim 38
    {8, 22, {1, 3, 1, 11, 11, 11, 77, 249}}, // dim 39
    {8, 38, {1, 3, 1, 11, 27, 43, 71, 9}}, // dim 40
    {8, 47, {1, 1, 7, 15, 21, 11, 81, 45}}, // dim 41
    {8, 49, {1, 3, 7, 3, 25, 31, 65, 79}}, // dim 42
    {8, 50, {1, 3, 1, 1, 19, 11, 3, 205}}, // dim 43
    {8, 52, {1, 1, 5, 9, 19, 21, 29, 157}}, // dim 44
    {8, 56, {1, 3, 7, 11, 1, 33, 89, 185}}, // dim 45
    {8, 67, {1, 3, 3, 3, 15, 9, 79, 71}}, // dim 46
    {8, 70, {1, 3, 7, 11, 15, 39, 119, 27}}, // dim 47
    {8, 84, {1, 1, 3, 1, 11, 31, 97, 225}}, // dim 48
    {8, 97, {1, 1, 1, 3, 23, 43, 57, 177}}, // dim 49
    {8, 103, {1, 3, 7, 7, 17, 17, 37, 71}}, // dim 50
    {8, 115, {1, 3, 1, 5, 27, 63, 123, 213}}, // dim 51
    {8, 122, {1, 1, 3, 5, 11, 43, 53, 133}}, // dim 52
    {9, 8, {1, 3, 5, 5, 29, 17, 47, 173, 479}}, // dim 53
    {9, 13, {1, 3, 3, 11, 3, 1, 109, 9, 69}}, // dim 54
    {9, 16, {1, 1, 1, 5, 17, 39, 23, 5, 343}}, // dim 55
    {9, 22, {1, 3, 1, 5, 25, 15, 31, 103, 499}}, // dim 56
    {9, 25, {1, 1, 1, 11, 11, 17, 63, 105, 183}}, // dim 57
    {9, 44, {1, 1, 5, 11, 9, 29, 97, 231, 363}}, // dim 58
    {9, 47, {1, 1, 5, 15, 19, 45, 41, 7, 383}}, // dim 59
    {9, 52, {1, 3, 7, 7, 31, 19, 83, 137, 221}}, // dim 60
    {9, 55, {1, 1, 1, 3, 23, 15, 111, 223, 83}}, // dim 61
    {9, 59, {1, 1, 5, 13, 31, 15, 55, 25, 161}}, // dim 62
    {9, 62, {1, 1, 3, 13, 25, 47, 39, 87, 257}}, // dim 63
};

// out[dim * 32 + bit] = direction number v_{bit+1} as a 32-bit fraction
inline void build_sobol_matrices(uint32_t out[64 * 32])
{
    for(int d = 0; d < 64; ++d)
    {
        uint32_t        m[33];
        const JoeKuoRow r = kJoeKuo[d];
        if(r.s == 0)
        {
            for(int k = 1; k <= 32; ++k) m[k] = 1;
        }
        else
        {
            for(int k = 1; k <= r.s && k <= 32; ++k) m[k] = r.m[k - 1];
            for(int k = r.s + 1; k <= 32; ++k)
            {
                uint32_t v = m[k - r.s] ^ (m[k - r.s] << r.s);
                for(int j = 1; j < r.s; ++j)
                    if((r.a >> (r.s - 1 - j)) & 1u) v ^= m[k - j] << j;
                m[k] = v;
            }
        }
        for(int k = 1; k <= 32; ++k) out[d * 32 + (k - 1)] = m[k] << (32 - k);
    }
}

}  // namespace prt
