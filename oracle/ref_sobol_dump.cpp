// TEST INFRASTRUCTURE (fixture generator, runs only in the build container).
// Links against the reference's own Sobol matrix table (/root/reference/external/NVIDIA_RTX/sobol/sobol.cpp,
// compiled where it lies; see oracle/Makefile target `_ref`) and writes
//   argv[1]: first 64 dimensions x 52 words as little-endian u32        (13 312 bytes)
//   argv[2]: text known-answer vectors produced by sobol::sample (sobol.h:40-55)
#include <cstdint>
#include <cstdio>
#include <sobol.h>
int main(int argc, char** argv)
{
    if(argc < 3) return 2;
    FILE* f = fopen(argv[1], "wb");
    if(!f) return 1;
    fwrite(sobol::Matrices::matrices, sizeof(uint32_t), 64 * sobol::Matrices::size, f);
    fclose(f);
    f = fopen(argv[2], "w");
    if(!f) return 1;
    fprintf(f, "# index dimension scramble float_hex float\n");
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for(int i = 0; i < 512; ++i)
    {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t index = (uint32_t)(s >> 32);
        if(i % 4 == 0) index >>= 12;
        if(i < 8) index = (uint32_t)i;
        uint32_t dim      = (uint32_t)(s & 63);
        uint32_t scramble = (uint32_t)((s >> 8) & 0xffffffffu);
        if(i % 7 == 0) scramble = 0;
        float v = sobol::sample(index, dim, scramble);
        uint32_t bits; __builtin_memcpy(&bits, &v, 4);
        fprintf(f, "%u %u %u 0x%08x %.9g\n", index, dim, scramble, bits, v);
    }
    fclose(f);
    return 0;
}
