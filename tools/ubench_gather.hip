// ubench_gather.hip — what does a divergent 16-byte gather cost a CU, and does it matter how the lanes of one
// instruction share cache lines? (measurement tool for DESIGN.md section 5; not part of the library)
//
//   A  every lane reads its own random 64-byte record with four 16-byte loads        (the BVH4 node visit)
//   B  quads of lanes read one record per instruction, 16 bytes per lane, four records per quad, four instructions
//      (same bytes, same instruction count, a quarter of the cache lines per instruction)
//   C  every lane reads 16 bytes of its own random record, one load                  (cost of one divergent instruction)
//   D  like A with 80-byte records and five loads                                      (the 8-wide node visit)
//   E  like B for 128-byte records: eight lanes per record, eight instructions
//   F  like A with aligned 128-byte records and eight loads                           (an 8-wide node with 16-bit planes)
//   G  like A, but 3 lanes in 8 read from the first 85 records only                    (lanes near the root of the tree)
//   H  like G with those lanes reading their record from a copy in LDS                 (top levels cached in LDS)
// Records come from a table of `tableMB` megabytes; indices are an LCG per lane, and each round's index depends on
// the data of the round before (a dependent chain per lane, as in a traversal).
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip ; run: tools/ubench_gather [tableMB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do                                                                                    \
    {                                                                                     \
        hipError_t e_ = (x);                                                              \
        if(e_ != hipSuccess)                                                              \
        {                                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while(0)

template <int MODE>
__global__ void __launch_bounds__(256, 6) k_gather(const uint4* __restrict__ table, uint32_t nrec, int rounds, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    __shared__ uint4 ldsTop[MODE == 7 ? 85 * 4 : 1];
    if(MODE == 7)
    {
        for(uint32_t k = threadIdx.x; k < 85u * 4u; k += 256u) ldsTop[k] = table[k];
        __syncthreads();
    }
    uint32_t  s    = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t  acc  = 0;
    for(int r = 0; r < rounds; ++r)
    {
        s                  = s * 1664525u + 1013904223u + (acc & 1u);  // depends on the last round's data
        const uint32_t idx = (uint32_t)(((unsigned long long)(s >> 4) * nrec) >> 28);
        if(MODE == 0)
        {
            const uint4* p = table + (size_t)idx * 4;
            const uint4  a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
        }
        else if(MODE == 1)
        {
            uint32_t v = 0;
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                const uint32_t ik = (uint32_t)__shfl((int)idx, (lane & ~3) + k, 64);
                const uint4    q  = table[(size_t)ik * 4 + (lane & 3)];
                v ^= q.x ^ q.w;
            }
            acc += v;
        }
        else if(MODE == 2)
        {
            const uint4 a = table[(size_t)idx * 4 + (lane & 3)];
            acc += a.x ^ a.w;
        }
        else if(MODE == 3)
        {
            const uint4* p = table + (size_t)idx * 5;
            const uint4  a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x;
        }
        else if(MODE == 6 || MODE == 7)
        {
            const bool     top = (lane & 7) < 3;
            const uint32_t it  = idx % 85u;
            if(MODE == 7 && top)
            {
                const uint4* p = ldsTop + it * 4;
                const uint4  a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z ^ d.w;
            }
            else
            {
                const uint4* p = table + (size_t)(top ? it : idx) * 4;
                const uint4  a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z ^ d.w;
            }
        }
        else if(MODE == 5)
        {
            const uint4* p = table + (size_t)idx * 8;
            const uint4  a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ h.w;
        }
        else
        {
            uint32_t v = 0;
#pragma unroll
            for(int k = 0; k < 8; ++k)
            {
                const uint32_t ik = (uint32_t)__shfl((int)idx, (lane & ~7) + k, 64);
                const uint4    q  = table[(size_t)ik * 8 + (lane & 7)];
                v ^= q.x ^ q.w;
            }
            acc += v;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, const uint4* table, size_t tableBytes, int recBytes, int loads, uint32_t* out, int blocks)
{
    const uint32_t nrec   = (uint32_t)(tableBytes / (MODE == 4 ? 128 : recBytes));
    const int      rounds = 2000;
    hipEvent_t     e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, nrec, 50, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, nrec, rounds, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves      = blocks * 4.0;
    const double waveInstr  = waves * rounds * loads;
    const double perCuNs    = ms * 1e6 / (waveInstr / 256.0);  // ns of one CU per wave-instruction
    const double bytes      = waveInstr * 64.0 * 16.0;
    printf("%-44s %8.3f ms  %7.1f ns/CU per wave-load  %6.2f lane-loads/clk/CU (2.4 GHz)  %7.1f GB/s\n", name, ms, perCuNs,
           64.0 / (perCuNs * 2.4), bytes / ms / 1e6);
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 4;
    const size_t tableBytes = mb << 20;
    uint4*       table = nullptr;
    uint32_t*    out   = nullptr;
    CHECK(hipMalloc(&table, tableBytes + 4096));
    const int blocks = 256 * 6;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<uint32_t> h(tableBytes / 4);
    uint32_t              s = 1;
    for(auto& v : h) v = (s = s * 1664525u + 1013904223u);
    CHECK(hipMemcpy(table, h.data(), tableBytes, hipMemcpyHostToDevice));
    printf("table %zu MB, %d blocks of 256 threads (6 waves per SIMD)\n", mb, blocks);
    if(argc > 2)  // one variant only (PMC calibration: records read = 1536 blocks x 256 lanes x 2050 rounds)
    {
        switch(argv[2][0])
        {
        case 'A': run<0>("A  lane: 4 x 16 B of its own 64-B record", table, tableBytes, 64, 4, out, blocks); break;
        case 'C': run<2>("C  lane: 16 B of its own record, 1 load", table, tableBytes, 64, 1, out, blocks); break;
        case 'D': run<3>("D  lane: 5 x 16 B of its own 80-B record", table, tableBytes, 80, 5, out, blocks); break;
        case 'F': run<5>("F  lane: 8 x 16 B of its own 128-B record", table, tableBytes, 128, 8, out, blocks); break;
        default: run<4>("E  oct: 8 records of 128 B, 16 B per lane", table, tableBytes, 128, 8, out, blocks); break;
        }
        return 0;
    }
    run<0>("A  lane: 4 x 16 B of its own 64-B record", table, tableBytes, 64, 4, out, blocks);
    run<1>("B  quad: 4 records, 16 B per lane per load", table, tableBytes, 64, 4, out, blocks);
    run<2>("C  lane: 16 B of its own record, 1 load", table, tableBytes, 64, 1, out, blocks);
    run<3>("D  lane: 5 x 16 B of its own 80-B record", table, tableBytes, 80, 5, out, blocks);
    run<4>("E  oct: 8 records of 128 B, 16 B per lane", table, tableBytes, 128, 8, out, blocks);
    run<5>("F  lane: 8 x 16 B of its own 128-B record", table, tableBytes, 128, 8, out, blocks);
    run<6>("G  like A, 3 lanes in 8 in the first 85 records", table, tableBytes, 64, 4, out, blocks);
    run<7>("H  like G, those lanes from an LDS copy", table, tableBytes, 64, 4, out, blocks);
    return 0;
}
