// prt_render — headless command-line driver: OBJ in, accumulation image out (PPM after pathRTpostprocess, PFM
// raw). Stands where the reference's main.cpp + vkContext::run stand (/root/reference/src/main.cpp:4-15), minus
// the window: the scene path is an argument instead of the hard-coded one (vkContext.cpp:52-72).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "Renderer.h"

using namespace prt_host;

int main(int argc, char** argv)
{
    if(argc < 2)
    {
        fprintf(stderr,
                "usage: %s scene.obj [--width W] [--height H] [--mode path|ao] [--spp N] [--bounces B] [--aa N]\n"
                "          [--eye x y z] [--target x y z] [--light x y z dx dy dz] [--light-area A] [--light-e E]\n"
                "          [--ao-rays N] [--ao-length L] [--device D] [--out image.ppm] [--pfm image.pfm]\n",
                argv[0]);
        return 2;
    }
    uint32_t    W = 1280, H = 720, spp = 64;
    int         device = 0;
    Settings    s;
    vec3        eye(0, 1, 3), target(0, 1, 0), lpos, ldir;
    bool        haveLight = false;
    std::string out = "out.ppm", pfm;
    for(int i = 2; i < argc; ++i)
    {
        auto f = [&](int k) { return (float)atof(argv[i + k]); };
        auto need = [&](int k) {
            if(i + k >= argc)
            {
                fprintf(stderr, "%s needs %d value(s)\n", argv[i], k);
                exit(2);
            }
        };
        if(!strcmp(argv[i], "--width")) { need(1); W = (uint32_t)atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--height")) { need(1); H = (uint32_t)atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--mode")) { need(1); s.rtRenderingMode = !strcmp(argv[++i], "ao") ? 1 : 0; }
        else if(!strcmp(argv[i], "--spp")) { need(1); spp = (uint32_t)atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--bounces")) { need(1); s.numIndicesBounces = atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--aa")) { need(1); s.numAArays = atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--eye")) { need(3); eye = vec3(f(1), f(2), f(3)); i += 3; }
        else if(!strcmp(argv[i], "--target")) { need(3); target = vec3(f(1), f(2), f(3)); i += 3; }
        else if(!strcmp(argv[i], "--light")) { need(6); lpos = vec3(f(1), f(2), f(3)); ldir = vec3(f(4), f(5), f(6)); haveLight = true; i += 6; }
        else if(!strcmp(argv[i], "--light-area")) { need(1); s.lightSourceArea = f(1); ++i; }
        else if(!strcmp(argv[i], "--light-e")) { need(1); s.lightE = f(1); ++i; }
        else if(!strcmp(argv[i], "--ao-rays")) { need(1); s.numAOrays = atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--ao-length")) { need(1); s.aoRayLength = f(1); ++i; }
        else if(!strcmp(argv[i], "--device")) { need(1); device = atoi(argv[++i]); }
        else if(!strcmp(argv[i], "--out")) { need(1); out = argv[++i]; }
        else if(!strcmp(argv[i], "--pfm")) { need(1); pfm = argv[++i]; }
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    try
    {
        std::vector<Model> models;
        models.emplace_back(std::string(argv[1]));
        if(!models[0].warnings.empty()) std::cout << models[0].warnings << "\n";
        Renderer r(device, W, H);
        s.samplesPerPixel = (int)spp;
        r.m_settings      = s;
        r.m_camera.setPosition(eye);
        r.m_camera.initDefaults((float)W / (float)H);
        r.m_camera.lookAtPoint(target);
        if(haveLight)
        {
            const vec3 d  = normalize(ldir);
            const vec3 up = std::fabs(d.y) < 0.99f ? vec3(0, 1, 0) : vec3(0, 0, -1);
            mat3       R;
            R[2] = -d;
            R[0] = normalize(cross(up, R[2]));
            R[1] = cross(R[2], R[0]);
            r.m_light.setOrientation(R);
            r.m_light.setPosition(lpos);
            r.m_light.setSize(vec2{s.lightSourceArea, s.lightSourceArea});
            r.m_light.setEmission(vec3(s.lightE, s.lightE, s.lightE));
            r.useAreaLight = true;
        }
        r.initRaytracing(models);
        r.render(1, s.rtRenderingMode == 1 ? 1u : spp, 1);
        const prt_stats st = r.stats();
        printf("%zu triangles, BVH %u nodes (%.1f ms), %llu rays, %llu iterations\n", models[0].m_indices.size() / 3,
               st.bvh_nodes, st.bvh_build_ms, (unsigned long long)st.rays, (unsigned long long)st.iterations);
        if(!out.empty() && !writePPM(out, r.postprocess(), W, H)) fprintf(stderr, "cannot write %s\n", out.c_str());
        if(!pfm.empty() && !writePFM(pfm, r.readAccum(), W, H)) fprintf(stderr, "cannot write %s\n", pfm.c_str());
    }
    catch(const std::exception& e)
    {
        std::cerr << e.what() << std::endl;  // main.cpp:7-15
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
