/* prt_host.h — C view of the headless host layer (libprt_host.so, sources in pathtracer_amd/host/).
 *
 * The host layer itself is C++ with the reference's scene API (Model, AreaLight, CameraControls, the
 * vkContext settings block; see the headers in pathtracer_amd/host/); these entry points exist so that tests, tools and
 * foreign-language callers can reach the same code through plain pointers and sizes. Each one names the
 * reference code it stands for. Functions return 0 on success, a negative value on failure.
 */
#ifndef PRT_HOST_H
#define PRT_HOST_H

#include "prt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prth_model prth_model;

/* VkTools::Model::Model(ctx, path) — /root/reference/src/Model.h:132-139: LoadModelFromFile + createTextures
 * (Model.cpp:47-221, 311-387). `err` receives the loader's message when the OBJ cannot be read. */
int  prth_model_load(const char* obj_path, prth_model** out, char* err, size_t err_len);
void prth_model_free(prth_model* m);
size_t              prth_model_num_vertices(const prth_model* m);  /* Model::m_vertices */
size_t              prth_model_num_indices(const prth_model* m);   /* Model::m_indices */
size_t              prth_model_num_materials(const prth_model* m); /* Model::m_materials */
size_t              prth_model_num_textures(const prth_model* m);  /* Model::m_textures */
const prt_vertex*   prth_model_vertices(const prth_model* m);
const uint32_t*     prth_model_indices(const prth_model* m);
const prt_material* prth_model_materials(const prth_model* m);
/* Model::m_textures[i] (RGBA8 pixels, size) and Model::m_texturePaths[i] ("" for the 1x1 white stand-in). */
int prth_model_texture(const prth_model* m, size_t i, prt_texture* out, const char** path);
const char* prth_model_warnings(const prth_model* m);

/* What the OBJ/MTL reader (tinyobj::LoadObj at Model.cpp:54) yields for a file, written as JSON. */
int prth_obj_dump_json(const char* obj_path, const char* mtl_dir, const char* json_path);

/* stbi_load(path, &w, &h, &channels, STBI_rgb_alpha) — Model.cpp:350. *rgba is released with prth_free. */
int  prth_image_load(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba);
void prth_free(void* p);

/* The reference's settings block (vkContext.h:161-192), as plain data. */
typedef struct prth_settings
{
    int32_t  numIndicesBounces, samplesPerPixel, numAArays;
    float    filterRadius;
    int32_t  numAOrays;
    float    aoRayLength, lightSourceArea, lightE, lightOtherE;
    int32_t  rtRenderingMode;
    uint32_t iteration;
} prth_settings;
void prth_default_settings(prth_settings* s);

/* CameraControls (CameraControls.cpp:35-41, 111-126) at a pose + vkContext::updateGraphicsUniforms
 * (vkContext.cpp:993-1060) with the light captured from the camera (m_moveLight): the 464-byte block, and the
 * un-flipped camera projection. yaw / pitch in degrees. */
int prth_camera_uniforms(float fov_deg, float aspect, float z_near, float z_far, const float position[3], float yaw_deg,
                         float pitch_deg, const prth_settings* settings, prt_uniforms* ubo_out, float projection_out[16]);

/* Whole path: Model -> Renderer::initRaytracing -> n iterations from `first_iteration` -> accumulation image
 * (width*height*4 floats). The camera sits at `eye` looking at `target`; the light is an AreaLight at `light_pos`
 * facing `light_dir` when light_pos != NULL, else it is captured from the camera. ubo_out (optional) receives
 * the block of the first iteration. Needs a GPU. */
int prth_render_obj(const char* obj_path, int device, uint32_t width, uint32_t height, const prth_settings* settings,
                    const float eye[3], const float target[3], const float* light_pos, const float* light_dir,
                    uint32_t first_iteration, uint32_t n_iterations, float* accum_out, prt_uniforms* ubo_out, char* err,
                    size_t err_len);

/* One entry of the top-level structure: vkRT_TLAS.cpp:58-72 AddInstance(blas, transform, instanceID,
 * hitGroupIndex). transform is a column-major mat4 (glm), object -> world. */
typedef struct prth_instance
{
    uint32_t modelIndex;
    float    transform[16];
    uint32_t instanceId;
    uint32_t hitGroupIndex;
} prth_instance;
/* Several models / instances flattened into one world-space model (pathtracer_amd/host/Scene.h); the reference
 * itself wires one identity instance of models[0] (vkRTX_setup.cpp:138-186). first_triangle (optional) receives
 * n_instances + 1 entries: gl_PrimitiveID base of every instance, then the total. */
int prth_flatten_instances(const prth_model* const* models, size_t n_models, const prth_instance* instances,
                           size_t n_instances, prth_model** out, uint32_t* first_triangle, char* err, size_t err_len);
/* The same models side by side in one object-space pool + one bottom-level range per model + the instance list: what
 * prt_set_scene + prt_set_instances take (Scene.h poolInstances). ranges: n_models entries, instances_out: n_instances. */
int prth_pool_instances(const prth_model* const* models, size_t n_models, const prth_instance* instances,
                        size_t n_instances, prth_model** out, prt_blas_range* ranges, prt_instance* instances_out, char* err,
                        size_t err_len);
/* Renderer::initRaytracing(models, instances, flatten) -> n iterations -> image (and, optionally, the primary
 * primitive ids of the last one). flatten = 0: two-level traversal on the GPU (one BVH per model); 1: the instances
 * copied into one world-space model. `moved` (optional, n_instances entries): transforms applied after the build
 * through Renderer::updateInstances (the top level's updateOnly path). Needs a GPU. */
int prth_render_instances(const prth_model* const* models, size_t n_models, const prth_instance* instances,
                          size_t n_instances, int flatten, int device, uint32_t width, uint32_t height,
                          const prth_settings* settings, const float eye[3], const float target[3], const float* light_pos,
                          const float* light_dir, uint32_t first_iteration, uint32_t n_iterations,
                          const prth_instance* moved, float* accum_out, uint32_t* hit_ids_out, char* err, size_t err_len);
/* Same as prth_render_obj for an already loaded (or flattened) model, with a camera-captured light or an
 * AreaLight; `resume_from` / `save_to` (optional paths) restore / store the progressive state around the n
 * iterations (Renderer::loadCheckpoint / saveCheckpoint). Needs a GPU. */
int prth_render_model(const prth_model* model, int device, uint32_t width, uint32_t height, const prth_settings* settings,
                      const float eye[3], const float target[3], const float* light_pos, const float* light_dir,
                      uint32_t first_iteration, uint32_t n_iterations, const char* resume_from, const char* save_to,
                      float* accum_out, char* err, size_t err_len);

/* vkContext::mainLoop / renderFrame (vkContext.cpp:92-231) without the window: n_frames calls of
 * Renderer::renderFrame with the reference's image semantics (PRT_FLAG_LITERAL), each preceded by the input events
 * scheduled for that frame. records[f] receives the parameter block of frame f (what vkContext memcpy's into the UBO,
 * vkContext.cpp:1057-1059), the progressive counter m_settings.iteration before and after updateGraphicsUniforms, and
 * whether the frame traced rays (vkContext.cpp:164: only while iteration < samplesPerPixel). Needs a GPU. */
#define PRTH_EVENT_MOVE_CAMERA 0 /* v = movement input (right, up, forward) for one second: CameraControls::updateMovements */
#define PRTH_EVENT_MOVE_LIGHT 1  /* the Space key: m_moveLight = true (the light jumps to the camera one frame later) */
#define PRTH_EVENT_SET_MODE 2    /* v[0] = 0: path tracing, 1: ambient occlusion (KP1 / KP2) */
typedef struct prth_frame_event
{
    uint32_t frame;
    int32_t  kind;
    float    v[3];
} prth_frame_event;
typedef struct prth_frame_record
{
    prt_uniforms ubo;
    uint32_t     settings_iteration_before, settings_iteration_after;
    int32_t      traced;
    uint32_t     pad_;
} prth_frame_record;
int prth_frame_loop(const prth_model* model, int device, uint32_t width, uint32_t height, const prth_settings* settings,
                    const float eye[3], const float target[3], const prth_frame_event* events, size_t n_events,
                    uint32_t n_frames, prth_frame_record* records, float* accum_out, char* err, size_t err_len);

/* Image and state files of the headless renderer (the reference only presents to a swapchain). */
int prth_write_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height);
int prth_checkpoint_write(const char* path, uint32_t width, uint32_t height, uint32_t next_iteration, int32_t mode,
                          int32_t bounces, int32_t aa_rays, const float* accum);
/* *accum is released with prth_free. */
int prth_checkpoint_read(const char* path, uint32_t* width, uint32_t* height, uint32_t* next_iteration, int32_t* mode,
                         int32_t* bounces, int32_t* aa_rays, float** accum, char* err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* PRT_HOST_H */
