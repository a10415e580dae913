// TEST INFRASTRUCTURE (CPU only). The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
// "ASan/UBSan for oracle + host"): `make -C oracle asan` compiles oracle.cpp together with this driver into
// oracle/_asan/oracle_selftest. The driver reads a scene and what the ordinary liboracle.so computed for it (files
// written by tests/test_oracle_cpu.py::test_oracle_under_sanitizers), runs the same calls — BVH build (single-level and
// per bottom-level structure), closest / any-hit traversal through the BVH and by brute force, AO and path renders
// (plain, literal, Russian roulette), post-process — and compares bit for bit. Any sanitizer report aborts (exit != 0).
//   oracle_selftest <dir> <width> <height> <threads>
#include "oracle.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::vector<unsigned char> slurp(const std::string& p, bool must = true)
{
    std::vector<unsigned char> b;
    FILE*                      f = fopen(p.c_str(), "rb");
    if(!f)
    {
        if(must) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(2); }
        return b;
    }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    b.resize((size_t)n);
    if(n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", p.c_str()); exit(2); }
    fclose(f);
    return b;
}
static int g_fail = 0;
static void same(const char* what, const void* got, const std::vector<unsigned char>& want)
{
    if(memcmp(got, want.data(), want.size()) != 0)
    {
        fprintf(stderr, "MISMATCH: %s\n", what);
        g_fail++;
    }
    else
        printf("ok  %s (%zu bytes)\n", what, want.size());
}

int main(int argc, char** argv)
{
    if(argc < 5) { fprintf(stderr, "usage: %s dir w h threads\n", argv[0]); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    const uint32_t    W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]);
    const int         threads = atoi(argv[4]);
    const auto verts = slurp(d + "verts.bin"), idx = slurp(d + "idx.bin"), mats = slurp(d + "mats.bin"),
               sobol = slurp(d + "sobol.bin"), rays = slurp(d + "rays.bin");
    const auto blas = slurp(d + "blas.bin", false), inst = slurp(d + "inst.bin", false);
    orc_scene* s = orc_scene_create((const prt_vertex*)verts.data(), verts.size() / sizeof(prt_vertex), (const uint32_t*)idx.data(),
                                    idx.size() / 4, (const prt_material*)mats.data(), mats.size() / sizeof(prt_material), nullptr, 0);
    if(!s) { fprintf(stderr, "orc_scene_create failed\n"); return 2; }
    uint32_t dims, size;
    memcpy(&dims, sobol.data(), 4);
    memcpy(&size, sobol.data() + 4, 4);
    orc_set_sobol(s, (const uint32_t*)(sobol.data() + 8), dims, size);
    orc_set_scramble_seed(s, 7);
    if(!inst.empty() &&
       orc_set_instances(s, (const prt_blas_range*)blas.data(), blas.size() / sizeof(prt_blas_range), (const prt_instance*)inst.data(),
                         inst.size() / sizeof(prt_instance)) != 0)
    {
        fprintf(stderr, "orc_set_instances failed\n");
        return 2;
    }
    if(orc_build_bvh(s) != 0) { fprintf(stderr, "orc_build_bvh failed\n"); return 2; }
    {
        const uint32_t        n = orc_bvh_num_nodes(s);
        std::vector<orc_node> nodes(n);
        std::vector<uint32_t> order(idx.size() / 3 + 1);
        if(inst.empty())
        {
            orc_bvh_read(s, nodes.data(), order.data());
            same("bvh nodes", nodes.data(), slurp(d + "expect_nodes.bin"));
        }
        else
            for(uint32_t k = 0; k < blas.size() / sizeof(prt_blas_range); ++k)
            {
                std::vector<orc_node> bn(orc_blas_num_nodes(s, k));
                orc_blas_read(s, k, bn.data(), order.data());
                same(("blas " + std::to_string(k) + " nodes").c_str(), bn.data(), slurp(d + "expect_blas" + std::to_string(k) + ".bin"));
            }
    }
    const size_t         nr = rays.size() / sizeof(orc_ray);
    std::vector<orc_hit> hits(nr);
    std::vector<uint8_t> occ(nr);
    orc_trace_closest(s, (const orc_ray*)rays.data(), nr, hits.data(), 0, threads);
    same("closest hits (BVH)", hits.data(), slurp(d + "expect_hits.bin"));
    orc_trace_closest(s, (const orc_ray*)rays.data(), nr, hits.data(), ORC_FLAG_BRUTE_FORCE, threads);
    same("closest hits (brute force)", hits.data(), slurp(d + "expect_hits.bin"));
    orc_trace_any(s, (const orc_ray*)rays.data(), nr, occ.data(), 0, threads);
    same("any-hit (BVH)", occ.data(), slurp(d + "expect_occ.bin"));
    orc_trace_any(s, (const orc_ray*)rays.data(), nr, occ.data(), ORC_FLAG_BRUTE_FORCE, threads);
    same("any-hit (brute force)", occ.data(), slurp(d + "expect_occ.bin"));
    struct Case
    {
        const char* name;
        int         mode;
        uint32_t    flags;
    } cases[] = {{"ao", 1, 0}, {"path", 0, 0}, {"path_literal", 0, ORC_FLAG_LITERAL}, {"path_rr", 0, ORC_FLAG_RUSSIAN_ROULETTE}};
    for(const Case& c : cases)
    {
        const auto ubo = slurp(d + "ubo_" + c.name + ".bin");
        if(ubo.size() != 2 * sizeof(prt_uniforms)) { fprintf(stderr, "bad ubo file\n"); return 2; }
        std::vector<float>    acc((size_t)W * H * 4, 0.0f);
        std::vector<uint32_t> ids((size_t)W * H);
        uint64_t              nrays = 0;
        for(int it = 0; it < 2; ++it)
            nrays += orc_render(s, (const prt_uniforms*)ubo.data() + it, W, H, c.mode, c.flags, acc.data(), ids.data(), threads);
        same((std::string("image ") + c.name).c_str(), acc.data(), slurp(d + "expect_img_" + c.name + ".bin"));
        same((std::string("ids ") + c.name).c_str(), ids.data(), slurp(d + "expect_ids_" + c.name + ".bin"));
        const auto wantRays = slurp(d + "expect_rays_" + c.name + ".bin");
        same((std::string("ray count ") + c.name).c_str(), &nrays, wantRays);
        if(c.mode == 0 && c.flags == 0)
        {
            std::vector<uint8_t> rgba((size_t)W * H * 4);
            orc_postprocess(acc.data(), W, H, rgba.data());
            same("post-process", rgba.data(), slurp(d + "expect_rgba.bin"));
        }
    }
    orc_scene_destroy(s);
    printf("%s\n", g_fail ? "FAILED" : "oracle selftest under ASan + UBSan: all equal");
    return g_fail ? 1 : 0;
}
