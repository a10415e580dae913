// Test helper: dumps the 64 x 32 Sobol words that the product generates (pathtracer_amd/csrc/sobol_table.hpp)
// so that a CPU-only test can compare them with the reference's table (tests/golden/sobol_64x52.u32).
#include <cstdio>

#include "../../pathtracer_amd/csrc/sobol_table.hpp"
int main()
{
    static uint32_t t[64 * 32];
    prt::build_sobol_matrices(t);
    fwrite(t, 4, 64 * 32, stdout);
    return 0;
}
