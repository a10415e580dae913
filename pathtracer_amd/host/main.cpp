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
                "          [--ao-rays N] [--ao-length L] [--device D] [--out image.ppm] [--pfm image.pfm] [--png image.png]\n"
                "          [--model more.obj]... [--instance MODEL x y z scale]... (default: one identity instance of model 0)\n"
                "          [--flatten]  (copy the instances into one world-space model instead of two-level traversal)\n"
                "          [--resume state.ckpt] [--checkpoint state.ckpt]  (continue / save the progressive render)\n",
                argv[0]);
        return 2;
    }
    uint32_t    W = 1280, H = 720, spp = 64;
    int         device = 0;
    Settings    s;
    vec3        eye(0, 1, 3), target(0, 1, 0), lpos, ldir;
    bool        haveLight = false, flatten = false;
    std::string out = "out.ppm", pfm, png, resume, checkpoint;
    std::vector<std::string> extraModels;
    std::vector<Instance>    instances;
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
        else if(!strcmp(argv[i], "--png")) { need(1); png = argv[++i]; }
        else if(!strcmp(argv[i], "--resume")) { need(1); resume = argv[++i]; }
        else if(!strcmp(argv[i], "--checkpoint")) { need(1); checkpoint = argv[++i]; }
        else if(!strcmp(argv[i], "--model")) { need(1); extraModels.push_back(argv[++i]); }
        else if(!strcmp(argv[i], "--flatten")) flatten = true;
        else if(!strcmp(argv[i], "--instance"))
        {
            need(5);
            Instance in;
            in.modelIndex = (uint32_t)atoi(argv[i + 1]);
            const float sc = f(5);
            in.transform[0][0] = in.transform[1][1] = in.transform[2][2] = sc;
            in.transform[3] = vec4(f(2), f(3), f(4), 1.0f);
            in.instanceId = (uint32_t)instances.size();
            instances.push_back(in);
            i += 5;
        }
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    try
    {
        std::vector<Model> models;
        models.emplace_back(std::string(argv[1]));
        if(!models[0].warnings.empty()) std::cout << models[0].warnings << "\n";
        for(const std::string& p : extraModels) models.emplace_back(p);
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
        size_t ntri = models[0].m_indices.size() / 3;
        if(instances.empty() && extraModels.empty())
            r.initRaytracing(models);
        else
        {
            if(instances.empty())  // every model once, where it stands
                for(size_t k = 0; k < models.size(); ++k)
                {
                    Instance in;
                    in.modelIndex = (uint32_t)k;
                    in.instanceId = (uint32_t)k;
                    instances.push_back(in);
                }
            ntri = 0;
            for(const Instance& in : instances)
                if(in.modelIndex < models.size()) ntri += models[in.modelIndex].m_indices.size() / 3;
            r.initRaytracing(models, instances, flatten);
        }
        uint32_t first = 1;
        if(!resume.empty())
        {
            r.loadCheckpoint(resume);
            first = r.m_settings.iteration;
        }
        const uint32_t n = s.rtRenderingMode == 1 ? 1u : spp;  // spp = iterations rendered by THIS run
        // one iteration consumes numAArays Sobol indices (sobolIndex++ per aaRay, pathRT.rgen:970), so the frame loop
        // advances by numAArays (vkContext.cpp:1033): the same stride here, or consecutive iterations would reuse samples
        const uint32_t stride = s.rtRenderingMode == 0 && s.numAArays > 1 ? (uint32_t)s.numAArays : 1u;
        r.m_settings.samplesPerPixel = (int)(first + n * stride);
        r.render(first, n, stride);
        r.m_settings.iteration = first + n * stride;
        if(!checkpoint.empty()) r.saveCheckpoint(checkpoint);
        const prt_stats st = r.stats();
        printf("%zu triangles, BVH %u nodes (%.1f ms), %llu rays, iterations %u..%u\n", ntri, st.bvh_nodes, st.bvh_build_ms,
               (unsigned long long)st.rays, first, first + (n - 1) * stride);
        if(!out.empty() && !writePPM(out, r.postprocess(), W, H)) fprintf(stderr, "cannot write %s\n", out.c_str());
        if(!pfm.empty() && !writePFM(pfm, r.readAccum(), W, H)) fprintf(stderr, "cannot write %s\n", pfm.c_str());
        if(!png.empty() && !writePNG(png, r.postprocess(), W, H)) fprintf(stderr, "cannot write %s\n", png.c_str());
    }
    catch(const std::exception& e)
    {
        std::cerr << e.what() << std::endl;  // main.cpp:7-15
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
