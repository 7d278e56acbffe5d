/*
 * rgk.h -- C ABI of the MI355X path-tracing core (librgk_hip.so).
 *
 * This is the drop-in boundary for the per-tile body of the reference's
 * RenderDriver::RenderRound (reference src/render_driver.cpp:144-190): the
 * lambda that builds a PathTracer (src/path_tracer.cpp:20-40), calls
 * Tracer::Render (src/tracer.cpp:6-37) on one RenderTask and merges the
 * private EXRTexture into the frame accumulator (src/texture.cpp:403-412).
 *
 * Plain C, plain pointers and sizes.  No C++ or torch types cross it.
 * Every entry point returns RGK_OK (0) or a negative rgk_status; nothing
 * throws across the ABI.  rgk_last_error() gives a thread-local message.
 *
 * All arithmetic on the path is float32 except the two plane dot products of
 * the triangle test (double, reference src/primitives.cpp:85,99-100).
 */
#ifndef RGK_H
#define RGK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rgk_status {
    RGK_OK = 0,
    RGK_ERR_INVALID = -1,   /* bad descriptor / argument                       */
    RGK_ERR_DEVICE = -2,    /* HIP runtime failure (message in rgk_last_error) */
    RGK_ERR_OOM = -3,       /* host or device allocation failed                */
    RGK_ERR_NO_DEVICE = -4, /* no HIP device visible                           */
    RGK_ERR_UNSUPPORTED = -5
} rgk_status;

/* ---- materials: mirrors Material + BxDF subclasses, src/bxdf/bxdf.hpp:19-159,
 *      brdf ids accepted by Material::LoadFromJson src/bxdf/bxdf.cpp:63-84 ---- */
typedef enum rgk_bxdf_kind {
    RGK_BXDF_DIFFUSE = 0,              /* "diffuse"/"diffusecosine"  bxdf.cpp:192-204 */
    RGK_BXDF_MIRROR = 1,               /* "mirror"                   bxdf.cpp:265-276 */
    RGK_BXDF_DIELECTRIC = 2,           /* "dielectric"               bxdf.cpp:332-408 */
    RGK_BXDF_TRANSPARENT = 3,          /* "transparent"              bxdf.cpp:412-423 */
    RGK_BXDF_MIX = 4,                  /* "mix"                      bxdf.cpp:235-249 */
    RGK_BXDF_LTC_BECKMANN = 5,         /* "ltc_beckmann"             bxdf.hpp:107-122 */
    RGK_BXDF_LTC_GGX = 6,              /* "ltc_ggx"                                   */
    RGK_BXDF_LTC_BECKMANN_DIFFUSE = 7, /* "ltc_beckmann_diffuse"     bxdf.hpp:125-159 */
    RGK_BXDF_LTC_GGX_DIFFUSE = 8       /* "ltc_ggx_diffuse"                           */
} rgk_bxdf_kind;

#define RGK_MAT_NO_RUSSIAN 1u /* Material::no_russian, bxdf.hpp:31 */

typedef struct rgk_material {
    uint32_t kind;       /* rgk_bxdf_kind */
    uint32_t flags;      /* RGK_MAT_* */
    float emission[3];   /* Material::emission */
    float roughness;     /* BxDFLTCBase::roughness */
    float ior;           /* BxDFDielectric::ior */
    float amount;        /* BxDFMix::amt1 */
    int32_t tex_diffuse; /* texture id or -1 (EmptyTexture: black, Empty()==true) */
    int32_t tex_color;   /* specular / mirror / dielectric colour texture, or -1 */
    int32_t tex_bump;    /* Material::bumpmap, or -1 */
    int32_t mix_m1;      /* BxDFMix::m1 material index, or -1 */
    int32_t mix_m2;      /* BxDFMix::m2 */
} rgk_material;

/* ---- textures: ReadableTexture family, src/texture.hpp:10-80 ---- */
typedef enum rgk_texture_kind {
    RGK_TEX_SOLID = 0,  /* SolidTexture: constant colour, slopes 0 */
    RGK_TEX_RGB32F = 1, /* FileTexture: width*height Color{r,g,b} float triples, row-major,
                           already gamma-decoded / flipped as the reference loader does -- what a binding to the reference hands
                           over (FileTexture::data).  One whose channels take <= 256 distinct values (every texture the reference
                           decodes from an 8-bit file, src/texture.cpp:203,252-254) is stored like RGK_TEX_RGB8 internally */
    RGK_TEX_RGB8 = 2    /* FileTexture decoded from an 8-bit file (PNG/JPEG, src/texture.cpp:189-292):
                           the bytes as stored (row-major, flipped as the loader does) plus the 256-entry
                           table that turns a byte into the float the reference would hold
                           (Color(byte/255).gammaDecode(2.2)).  Same texel values as RGK_TEX_RGB32F at a
                           quarter of the memory traffic. */
} rgk_texture_kind;

typedef struct rgk_texture {
    uint32_t kind;
    uint32_t width, height;
    float color[3];         /* RGK_TEX_SOLID */
    const float *texels;    /* RGK_TEX_RGB32F: 3*width*height floats */
    const uint8_t *texels8; /* RGK_TEX_RGB8: 3*width*height bytes */
    const float *lut;       /* RGK_TEX_RGB8: 256 floats, byte -> channel value */
} rgk_texture;

/* ---- point / sphere lights: Light{FULL_SPHERE}, src/primitives.hpp:26-43,
 *      filled by ConfigJSON::InstallLights src/config.cpp:372-387 ---- */
typedef struct rgk_pointlight {
    float pos[3];
    float color[3];
    float intensity;
    float size;
} rgk_pointlight;

typedef enum rgk_sky_mode { RGK_SKY_COLOR = 0, RGK_SKY_ENVMAP = 1 } rgk_sky_mode;

/* ---- the immutable Scene as it crosses the seam (src/scene.hpp:76-172) ---- */
typedef struct rgk_scene_desc {
    uint32_t n_vertices;
    const float *vertices;  /* 3*n  Scene::vertices  */
    const float *normals;   /* 3*n  Scene::normals   */
    const float *tangents;  /* 3*n  Scene::tangents  */
    const float *texcoords; /* 2*n  Scene::texcoords */

    uint32_t n_triangles;
    const uint32_t *tri_indices;  /* 3*n  Triangle::va,vb,vc */
    const uint32_t *tri_material; /* n    Triangle::mat as an index */

    uint32_t n_materials;
    const rgk_material *materials;
    uint32_t n_textures;
    const rgk_texture *textures;

    uint32_t n_pointlights;
    const rgk_pointlight *pointlights;

    /* Scene::areal_lights (src/scene.hpp:91-103): one group per emissive
     * primitive / mesh, listing its triangle ids in insertion order.       */
    uint32_t n_areal_lights;
    const uint32_t *areal_offsets; /* n_areal_lights+1 */
    const uint32_t *areal_tris;

    /* sky, Scene::SetSkyboxColor / SetSkyboxEnvmap src/scene.hpp:122-133 */
    uint32_t sky_mode;
    float sky_color[3];
    float sky_intensity;
    float sky_rotate;
    int32_t sky_texture;

    /* LTC fits (src/LTC/ltc_ggx.cpp, ltc_beckmann.cpp): 64*64 entries of
     * {m0,m2,m4,m6,amplitude} as float32 (the only non-constant entries of
     * tabM; m8==1, the rest 0).  Either may be NULL if no material uses it. */
    const float *ltc_ggx;
    const float *ltc_beckmann;

    /* RGK_BUILD_*: how the accelerator is built (results never depend on it: hits come from the triangle records) */
    uint32_t build_flags;
} rgk_scene_desc;

#define RGK_BUILD_AUTO 0u     /* default: the host builder below half a million triangle references, the device builder from there on */
#define RGK_BUILD_HOST_SAH 4u /* binned SAH + reinsertion on the host, collapsed to the 4-wide quantised BVH: the best traversal, seconds
                                 to build at a million triangles                                                              */
#define RGK_BUILD_KEEP_FLOAT_TEXTURES 2u /* store every RGK_TEX_RGB32F texture as float4 texels, even one with <= 256 distinct channel
                                            values (by default such a texture is stored as bytes + the table of its values: the same
                                            texel values, a quarter of the traffic)                                                */
#define RGK_BUILD_DEVICE 1u   /* LBVH on the GPU (Morton sort, Karras hierarchy, refit with tree rotations, collapse + quantisation: all on
                                 the device): a tenth of the host's build time, traversal within a few per cent of it        */

/* ---- Camera as RenderRound(const Camera&) receives it (src/render_driver.cpp:146): the public data members
 *      of the reference's Camera, src/camera.hpp:27-41, under their own names (`lookat` is not read on the path).
 *      A host that holds a Camera copies the members; a host that only has the constructor arguments
 *      (ConfigJSON::GetCamera, src/config.cpp:332-370) calls rgk_camera_init below. ---- */
typedef struct rgk_camera {
    float origin[3];
    float direction[3];
    float cameraup[3];
    float cameraleft[3];
    float viewscreen[3];
    float viewscreen_x[3];
    float viewscreen_y[3];
    float lens_size;
    int32_t xsize, ysize;
} rgk_camera;

typedef enum rgk_sampler_kind {
    RGK_SAMPLER_HALTON = 0,    /* counter-based Faure-Halton + Cranley-Patterson */
    RGK_SAMPLER_STRATIFIED = 1 /* oracle only: reference StratifiedSampler, src/sampler.cpp:85-116 */
} rgk_sampler_kind;

#define RGK_FLAG_COUNT_TRAVERSAL 1u /* fill node_visits / tri_tests (slower kernels) */
#define RGK_FLAG_TIME_KERNELS 2u    /* bracket every kernel class with HIP events     */

/* ---- PathTracer constructor arguments, src/path_tracer.cpp:20-40 ---- */
typedef struct rgk_params {
    uint32_t xres, yres;
    uint32_t multisample;
    uint32_t depth;
    float clamp;
    float russian;
    float bumpmap_scale;
    uint32_t force_fresnell; /* stored, unused -- as in the reference */
    uint32_t reverse;
    uint32_t sampler; /* rgk_sampler_kind */
    uint32_t flags;   /* RGK_FLAG_* */
} rgk_params;

/* ---- RenderTask (src/tracer.hpp:14-24) + its PathTracer seed
 *      (seedstart + c, src/render_driver.cpp:160,173) ---- */
typedef struct rgk_tile {
    uint32_t x0, x1, y0, y1; /* xrange_start, xrange_end, yrange_start, yrange_end */
    uint32_t seed;
} rgk_tile;

/* The kernels of a round, for the per-kernel records in rgk_counters: RGK_FLAG_TIME_KERNELS / RGK_FLAG_COUNT_TRAVERSAL. */
typedef enum rgk_kernel_id {
    RGK_K_TRACE_CAMERA = 0, /* k_trace_camera: bounce 0, camera rays made where they are traced          */
    RGK_K_TRACE_CLOSEST,    /* k_trace_closest: camera-path bounces >= 1                                 */
    RGK_K_SHADE_FIRST,      /* k_shade<., FIRST = true>: the first vertex of every path                  */
    RGK_K_SHADE,            /* k_shade<., FIRST = false>: later vertices                                 */
    RGK_K_SHADOW_FIRST,     /* k_trace_shadow_first: first-vertex NEE rays from light-side entry nodes   */
    RGK_K_SHADOW,           /* k_trace_shadow: NEE rays                                                  */
    RGK_K_SHADOW_JOBS,      /* k_trace_shadow_jobs: vertices with connections (reverse > 0)              */
    RGK_K_CONNECT,          /* k_connect (reverse > 0)                                                   */
    RGK_K_LIGHT_TRACE,      /* k_trace_closest of the light sub-path (reverse > 0)                       */
    RGK_K_LIGHT_SHADE,      /* k_raygen_light + k_list_hits + k_shade_light                              */
    RGK_K_LIGHT_SPLAT,      /* k_trace_shadow in splat mode: light-tracing side effects                  */
    RGK_K_RESOLVE,          /* k_resolve[_tiled]                                                         */
    RGK_K_OTHER,            /* per-round / per-frame lists, counters, progress marks                     */
    RGK_K_COUNT
} rgk_kernel_id;

typedef struct rgk_kernel_stat {
    double ms;            /* RGK_FLAG_TIME_KERNELS: summed HIP-event time on the scene's stream           */
    uint32_t launches;
    uint32_t reserved;
    uint64_t units;       /* rays (trace kernels) / vertices (shade, connect, jobs) / paths (resolve)     */
    uint64_t node_visits; /* RGK_FLAG_COUNT_TRAVERSAL, trace kernels                                      */
    uint64_t tri_tests;
} rgk_kernel_stat;

typedef struct rgk_counters {
    uint64_t paths;        /* pixels * multisample processed                      */
    uint64_t path_rays;    /* reference semantics: raycount++ path_tracer.cpp:126 */
    uint64_t shadow_rays;  /* Visibility() calls                                  */
    uint64_t node_visits;  /* closest-hit traversal, RGK_FLAG_COUNT_TRAVERSAL     */
    uint64_t tri_tests;
    uint64_t shadow_node_visits;
    uint64_t shadow_tri_tests;
    double ms_trace;       /* RGK_FLAG_TIME_KERNELS: summed HIP-event time, ms    */
    double ms_shadow;
    double ms_shade;
    double ms_other;
    uint32_t n_trace_launches;
    uint32_t n_shadow_launches;
    uint32_t n_shade_launches;
    uint32_t reserved;
    rgk_kernel_stat kernel[RGK_K_COUNT]; /* the same, per kernel (the four class sums above are sums of these) */
} rgk_counters;

typedef struct rgk_scene_info {
    float epsilon;        /* Scene::epsilon, src/scene.cpp:390   */
    float bbox_min[3];    /* xBB/yBB/zBB .first,  scene.cpp:393  */
    float bbox_max[3];
    float total_areal_power, total_point_power; /* scene.cpp:323-344 */
    uint32_t n_nodes;     /* accelerator nodes                    */
    uint32_t node_bytes;  /* s_node of SURVEY 8(d)                */
    uint32_t tri_bytes;   /* s_tri                                */
    uint32_t max_depth;
    uint32_t n_leaf_refs;
    uint32_t n_float_textures;      /* RGK_TEX_RGB32F textures handed over ...                                          */
    uint32_t n_palettized_textures; /* ... and how many of them are stored as bytes + value table (<= 256 distinct values) */
} rgk_scene_info;

typedef struct rgk_hit {
    float t;
    int32_t tri; /* triangle index or -1 */
    float a, b, c; /* Intersection::a,b,c  (c'=1-alpha-beta, alpha, beta) scene_intersect.cpp:280-283 */
} rgk_hit;

/* Progress of the round a scene is rendering, for a monitor thread (the reference's FrameMonitorThread reads pixels_done /
 * rays_done every 100 ms, src/render_driver.cpp:49-139).  Fed from the device's side of the stream: a host callback behind every
 * bounce of every pass bumps `stage`.  pixels_done of the reference = round_pixels * stage / stages (+ round_pixels per finished round). */
typedef struct rgk_progress {
    uint32_t stage, stages;      /* bounces finished / bounces in the round (over all passes); stage == stages: round done */
    uint32_t rounds;             /* rounds this scene has finished since it was created                                    */
    uint32_t busy;               /* 1 while a round is in flight                                                            */
    uint64_t round_pixels;       /* pixels of the round in flight (or of the last one)                                      */
    uint64_t round_paths;        /* pixels * multisample                                                                     */
} rgk_progress;

typedef struct rgk_scene rgk_scene;

const char *rgk_last_error(void);
int rgk_device_count(void);

/* Scene::Commit() outputs (src/scene.cpp:294-400) + accelerator build + upload. */
int rgk_scene_create(const rgk_scene_desc *desc, int device, rgk_scene **out);
/* Moved vertices, same triangles (an animated mesh between two frames; SURVEY 8(f) f2 "refit").  What Scene::Commit derives
 * from the positions is recomputed -- triangle planes, epsilon = 1e-5 * diagonal, the padded box, the areal-light tables
 * (src/scene.cpp:294-400) -- and the accelerator is REFIT on the device: its topology stays, every box is recomputed bottom-up
 * (milliseconds at a million triangles, where a rebuild takes 0.15 - 1.5 s).  Hits are those of a freshly created scene with
 * the same vertices (they come from the triangle records; boxes only steer the walk); the walk is as good as the old topology
 * fits the new positions.  vertices: 3 * n_vertices floats; normals / tangents: the same, or NULL = unchanged.  A triangle
 * that was degenerate at creation is not in the tree and stays out.  Not while a round is in flight on this scene. */
int rgk_scene_refit(rgk_scene *scene, const float *vertices, const float *normals, const float *tangents);
void rgk_scene_destroy(rgk_scene *scene);
int rgk_scene_get_info(const rgk_scene *scene, rgk_scene_info *out);

/* Callable from ANY thread while another thread is inside rgk_render_round*: never blocks, touches no device state. */
int rgk_scene_get_progress(const rgk_scene *scene, rgk_progress *out);

/* Tuning switches of ONE scene; none of them changes a result (the tests render with each and compare bits).  Keys:
 * "entry_points", "entry_cap", "light_entry" (0 / 1: where camera rays and first shadow rays start their walk), "sample_group"
 * (log2 of the samples of a pixel that sit side by side in the path-slot order, 0..6; -1: default), "batch_paths" (paths per
 * pass; 0: sized from the free memory), "workspace_gb" (0: default), "beam" (0 / 1: camera rays of a pixel walk the tree together), "two_lanes" (0 / 1: an experiment, off).  Their initial values are read from the environment
 * (RGK_ENTRY_POINTS, RGK_ENTRY_CAP, RGK_LIGHT_ENTRY, RGK_SAMPLE_GROUP, RGK_BATCH_PATHS, RGK_WORKSPACE_GB) ONCE, in
 * rgk_scene_create; a round never reads the environment.  Not while a round is in flight on this scene. */
int rgk_scene_set_tuning(rgk_scene *scene, const char *key, double value);

/* GenerateTaskList, src/render_driver.cpp:30-46.  Tiles of tile_size, sorted by
 * distance of the tile midpoint to (mid_x, mid_y); ties broken by (y0, x0)
 * (the reference's std::sort leaves ties unspecified).  `seed` of
 * tile i is seedstart + seedcount_base + i.  Call with tiles==NULL for the count. */
int rgk_generate_task_list(uint32_t tile_size, uint32_t xres, uint32_t yres, float mid_x,
                           float mid_y, uint32_t seedstart, uint32_t seedcount_base,
                           rgk_tile *tiles, uint32_t *n_tiles);

/* Camera::Camera(pos, la, up, yview, xview, xsize, ysize, focus_plane, lens_size), src/camera.cpp:7-24: fills the
 * derived members from the constructor arguments, in the reference's operation order (host only, no device needed). */
int rgk_camera_init(rgk_camera *out, const float pos[3], const float lookat[3], const float up[3], float yview,
                    float xview, int32_t xsize, int32_t ysize, float focus_plane, float lens_size);

/* The per-task body of RenderRound for a list of tiles (render_driver.cpp:158-184):
 * accum_rgb[3*(y*xres+x)..] += sum over samples, accum_count[y*xres+x] += multisample.
 * Host buffers; blocking. */
int rgk_render_round(rgk_scene *scene, const rgk_camera *camera, const rgk_params *params,
                     const rgk_tile *tiles, uint32_t n_tiles, float *accum_rgb,
                     uint32_t *accum_count, rgk_counters *counters);

/* Same, adding into DEVICE buffers on the scene's GPU (the per-GPU private
 * accumulator that RCCL then reduces).  Blocking unless the stream is the caller's. */
int rgk_render_round_device(rgk_scene *scene, const rgk_camera *camera,
                            const rgk_params *params, const rgk_tile *tiles, uint32_t n_tiles,
                            float *d_accum_rgb, uint32_t *d_accum_count,
                            rgk_counters *counters);

/* ---- the frame accumulator on the device: EXRTexture total_ob of RenderFrame (src/render_driver.cpp:199,
 *      src/texture.hpp:83-118: `data` = sum of radiance per pixel, `count` = samples per pixel).  A host without HIP
 *      headers keeps its frame here and hands rgk_accum_rgb / rgk_accum_count to rgk_render_round_device. ---- */
typedef struct rgk_accum rgk_accum;
int rgk_accum_create(uint32_t xres, uint32_t yres, int device, rgk_accum **out); /* zeroed */
void rgk_accum_destroy(rgk_accum *acc);
int rgk_accum_clear(rgk_accum *acc);
float *rgk_accum_rgb(rgk_accum *acc);      /* DEVICE pointer, 3 * xres * yres floats */
uint32_t *rgk_accum_count(rgk_accum *acc); /* DEVICE pointer, xres * yres */
int rgk_accum_download(const rgk_accum *acc, float *rgb, uint32_t *count); /* to host buffers (either may be NULL) */
int rgk_accum_upload(rgk_accum *acc, const float *rgb, const uint32_t *count);
/* dst += src (both on the same device): EXRTexture::Accumulate (src/texture.cpp:403-412).  What the root rank does with a
 * round's reduced sum -- see rgk_accum_reduce. */
int rgk_accum_add(rgk_accum *dst, const rgk_accum *src);
/* A digest of what the frame is rendered from (scene, camera, parameters; the host's choice of hash, 0 = none).  rgk_accum_save
 * stores it, rgk_accum_load refuses a file whose tag differs from the accumulator's (either side untagged: not compared). */
int rgk_accum_set_tag(rgk_accum *acc, uint64_t tag);

/* Raw-accumulator checkpoint (the reference cannot resume a frame): the accumulator with the two numbers
 * that make the next round continue the sequence -- rounds done and the running task counter `seedcount`
 * (src/render_driver.cpp:160,222).  Rendering k rounds, saving, loading and rendering m more gives the bits of k + m rounds. */
int rgk_accum_save(const rgk_accum *acc, const char *path, uint32_t rounds_done, uint32_t seedcount);
int rgk_accum_load(rgk_accum *acc, const char *path, uint32_t *rounds_done, uint32_t *seedcount);

/* ---- multi-GPU: one process per GPU, tiles dealt round-robin, ONE exchange per round ---- */
typedef struct rgk_comm rgk_comm;
#define RGK_COMM_ID_BYTES 128
/* Tile i of the centre-out list goes to rank i mod world_size (tiles keep their seeds: the image does not depend on
 * the number of GPUs).  Call with out == NULL for the count. */
int rgk_shard_tiles(const rgk_tile *tiles, uint32_t n_tiles, int rank, int world_size, rgk_tile *out, uint32_t *n_out);
/* RCCL communicator over the node's xGMI links.  Rank 0 makes the id and hands its 128 bytes to the other ranks by the
 * host's own means (a file, an environment variable, MPI, ...).  librccl is bound at run time: single-GPU hosts need none. */
int rgk_comm_get_unique_id(uint8_t id[RGK_COMM_ID_BYTES]);
int rgk_comm_create(const uint8_t id[RGK_COMM_ID_BYTES], int rank, int world_size, int device, rgk_comm **out);
void rgk_comm_destroy(rgk_comm *comm);
/* total_ob.Accumulate(output_buffer) under total_ob_mx (src/render_driver.cpp:177-182) across GPUs: in-place sum-reduce of
 * the per-GPU DEVICE accumulators to `root` (d_accum_count may be NULL when the host derives counts analytically).
 * Collective and blocking: every rank calls it once per round.  IN PLACE means: afterwards the root's buffer holds the sum
 * over all ranks and every other rank's buffer still holds its own contribution -- so what is reduced must be ONE ROUND's
 * accumulator (rgk_accum_clear before the round), which the root then adds to its frame total (rgk_accum_add).  Reducing a
 * buffer that keeps growing over the rounds would count the other ranks' earlier rounds again with every reduce. */
int rgk_accum_reduce(rgk_comm *comm, float *d_accum_rgb, uint32_t *d_accum_count, uint32_t xres, uint32_t yres, int root);

/* Scene::FindIntersectKdOtherThan (src/scene_intersect.cpp:211-327) for n rays.
 * rays: 8 floats each {ox,oy,oz, dx,dy,dz, near, far}; ignore: triangle id or -1. */
int rgk_trace_closest(rgk_scene *scene, uint32_t n, const float *rays, const int32_t *ignore,
                      rgk_hit *hits, rgk_counters *counters);

/* Scene::Visibility (src/scene.cpp:670-673) for n point pairs (3 floats each). */
int rgk_trace_visibility(rgk_scene *scene, uint32_t n, const float *a, const float *b,
                         uint8_t *visible, rgk_counters *counters);

/* BxDF::value(Vi, Vr, texUV) and BxDF::sample(Vi, texUV, sample) (src/bxdf/bxdf.hpp:38-43) of material mat[i], for n
 * inputs, in the local frame (+Z = shading normal).  Unit-level seams for the parity tests.  route 0: the generic code (every
 * material kind, mixes included); route 1: what the shading kernel runs for diffuse / LTC materials (texture colours and LTC
 * table entries fetched once and shared between value and sample).  Vi, Vr, out_*: 3 floats each; uv, u: 2 floats each. */
int rgk_bxdf_value(rgk_scene *scene, uint32_t n, uint32_t route, const uint32_t *mat, const float *Vi, const float *Vr,
                   const float *uv, float *out_rgb);
int rgk_bxdf_sample(rgk_scene *scene, uint32_t n, uint32_t route, const uint32_t *mat, const float *Vi, const float *uv,
                    const float *u, float *out_dir, float *out_weight, uint8_t *may_leak);
/* ReadableTexture::GetPixelInterpolated, GetSlopeRight, GetSlopeBottom (src/texture.cpp:35-102, src/texture.hpp:64-80) of
 * descriptor texture tex[i] (-1: EmptyTexture) at uv[i]. */
int rgk_texture_sample(rgk_scene *scene, uint32_t n, const int32_t *tex, const float *uv, float *rgb, float *slope_right,
                       float *slope_bottom);

/* The pinned transcendental functions of the path (include/rgk_libm.h) evaluated on the device, for the test that the GPU
 * and the CPU produce the same bits: fn 0 sin, 1 cos, 2 acos, 3 asin, 4 atan2(a[i], b[i]) (b may be NULL otherwise). */
int rgk_libm_eval(int fn, uint32_t n, const float *a, const float *b, float *out);

/* Sampler::Get1D / Get2D of the build's Halton sampler evaluated on the device:
 * out[2*i..] = sample of (seed[i], index[i], dim[i]); is2d selects Get2D. */
int rgk_sampler_eval(uint32_t n, const uint32_t *seed, const uint32_t *index,
                     const uint32_t *dim, int is2d, float *out);

/* ---- output path: what RenderFrame does with total_ob after each round ---------------------- */

/* EXRTexture::Normalize followed by the GetPixel that Write applies (src/texture.cpp:349-354,376-400,
 * src/render_driver.cpp:233,245): out_rgb[p] = (accum_rgb[p] * val) / accum_count[p], 0 where the count is 0.
 * output_scale <= 0 ("auto", the reference's default, config.cpp:303-311): val = 1 / (largest channel of
 * accum/count), so the brightest channel becomes 1.  Host buffers; *scale_used receives val (may be NULL). */
int rgk_output_normalize(const float *accum_rgb, const uint32_t *accum_count, uint32_t xres, uint32_t yres,
                         float output_scale, float *out_rgb, float *scale_used);

/* EXRTexture::Write (src/texture.cpp:356-374): a scan-line OpenEXR file of half-float R, G, B from rgb
 * (xres * yres * 3 floats, row-major from the top row) and A = 1.  float -> half is OpenEXR's conversion (round
 * to nearest even, overflow to infinity).  Written uncompressed; the reference's RgbaOutputFile default is PIZ --
 * the same pixels in another container encoding. */
int rgk_output_write_exr(const char *path, uint32_t xres, uint32_t yres, const float *rgb);

/* float -> IEEE half bits exactly as the writer stores them (exposed for the tests). */
uint16_t rgk_float_to_half(float v);

#ifdef __cplusplus
}
#endif
#endif /* RGK_H */
