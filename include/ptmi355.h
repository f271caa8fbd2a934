/*
 * ptmi355.h -- C ABI of libptmi355.so, the MI355X-native (gfx950 / HIP) replacement for the
 * per-iteration render path of CIS565 Project2-Pathtracer.
 *
 * The reference's host code reaches this path through ONE C++ symbol,
 *     void cudaRaytraceCore(uchar4*, camera*, int frame, int iterations,
 *                           material*, int, geom*, int);
 * (declared /root/reference/src/raytraceKernel.h:18, defined src/raytraceKernel.cu:164-227,
 * called src/main.cpp:126).  The adaptor TU project2-pathtracer_amd/adaptor/cuda_raytrace_core.cpp
 * defines exactly that symbol and forwards to the functions below; nothing in this header
 * mentions a C++ or a torch type: plain pointers, sizes and POD structs only.
 *
 * Every function returns 0 on success and a negative pt_status otherwise; pt_last_error()
 * returns the message.  The reference's convention (print "Cuda error: ..." and
 * exit(EXIT_FAILURE), src/raytraceKernel.cu:20-26) is applied by the adaptor, not here.
 *
 * There is NO CPU fallback: pt_create fails with PT_ERR_NO_DEVICE when no gfx950 device is
 * visible, and every render entry point needs a context.
 */
#ifndef PTMI355_H
#define PTMI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: pt_config grew by `streams` + reserved[3] (1 ended at `direct_light`); always fill pt_config through
 * pt_config_default() first, so that fields added later keep their defaults
 * 3: + pt_get_rows, pt_gather_rows_peer (the per-frame exchange of a row-sharded render)
 * 4: + pt_mesh, pt_set_meshes, pt_scene_mesh_count, pt_scene_mesh (GEOMTYPE MESH)
 * 5: ordering 2 / 3 and bvh 1 / 2 (round-1 experiments, all slower than what replaced them) are gone; the fields stay
 * 6: ordering = 2 is back with a new meaning: whole paths on the typed work queues, one launch per group
 * 7: geometry_path, compaction = 1 (look-back scan) and merge_floor are gone (measured slower; the fields stay and
 *    must be 0); reserved[3] became cluster_size / path_static_eighths / wide_variant (they replace environment
 *    switches the library used to read); ordering = 2 also covers scenes of 33..256 analytic primitives (k_path_w)
 * 8: + pt_config.grid_density (k_path_w walks a uniform grid over the small primitives instead of testing every cluster box)
 * 9: pt_config loses the fields that had to be 0 (geometry_path, compaction, merge_floor, bvh) and the A/B switches whose best
 *    setting is built in (cluster_size, path_static_eighths, wide_variant): 18 fields; pt_config_default now gives the fast path
 *    (ordering = 2, streams = 2); ordering = 2 takes any number of analytic primitives; + pt_debug_fan_probe */
#define PTMI355_ABI_VERSION 9

typedef enum {
    PT_OK = 0,
    PT_ERR_NO_DEVICE = -1,      /* no HIP device / not gfx950 */
    PT_ERR_HIP = -2,            /* a HIP runtime call failed (message has the call) */
    PT_ERR_ARGUMENT = -3,
    PT_ERR_STATE = -4,          /* e.g. render before pt_upload_scene */
    PT_ERR_PARSE = -5,          /* scene file */
    PT_ERR_IO = -6
} pt_status;

/* == `material`, /root/reference/src/sceneStructs.h:62-73 (64 bytes, same field order) */
typedef struct {
    float color[3];
    float specularExponent;
    float specularColor[3];
    float hasReflective;
    float hasRefractive;
    float indexOfRefraction;
    float hasScatter;
    float absorptionCoefficient[3];
    float reducedScatterCoefficient;
    float emittance;
} pt_material;

/* The part of `staticGeom` (src/sceneStructs.h:32-40) the kernels read, per frame: type
 * (GEOMTYPE: 0 SPHERE, 1 CUBE, 2 MESH -- src/sceneStructs.h:14), materialid, and rows x,y,z
 * of transform / inverseTransform (cudaMat4 rows, src/cudaMat4.h:18-23; the w row is
 * (0,0,0,1) for every matrix scene.cpp builds and multiplyMV never reads it,
 * src/intersections.h:53-59).  104 bytes. */
typedef struct {
    int   type;
    int   materialid;
    float transform[12];
    float inverseTransform[12];
} pt_geom;

/* == `cameraData`, src/sceneStructs.h:42-48 (52 bytes); fov = half-angles in degrees */
typedef struct {
    float resolution[2];
    float position[3];
    float view[3];
    float up[3];
    float fov[2];
} pt_camera;

/* Render options the reference hard-codes or lacks (traceDepth src/raytraceKernel.cu:166;
 * README.md:47-51,63).  Always fill through pt_config_default() first, then change what differs: fields added later keep their
 * defaults.  pt_config_default gives the fast path: whole paths in one launch per group of iterations (ordering = 2) on two
 * streams per GPU (streams = 2), 8 bounces, all rows. */
typedef struct {
    int   device;            /* HIP device ordinal */
    int   mode;              /* 0 = path trace (generate -> bounce* -> accumulate)
                                1 = the reference kernel as shipped: one hit, flat material
                                    colour OVERWRITES the pixel (src/raytraceKernel.cu:123-159) */
    int   max_depth;         /* bounces per path, 1..64 (mode 0) */
    int   camera_mode;       /* 0 = reference ray incl. normalize(R) (raytraceKernel.cu:67-69)
                                1 = corrected pinhole / thin lens */
    int   antialias;         /* jitter pixel position by U(-.5,.5)^2 */
    float aperture;          /* thin-lens radius, camera_mode 1; 0 = pinhole */
    float focal_distance;
    int   row_offset;        /* multi-GPU: this context owns rows y with y % row_stride == */
    int   row_stride;        /*   row_offset (single GPU: 0, 1) */
    int   chunk_rays;        /* slots per pool segment (0 = by launch size) / camera rays per job of the whole-path kernels */
    int   blocks_per_cu;     /* persistent grid size = CUs * this (0 = default) */
    int   profile;           /* 1 = bracket every kernel launch with HIP events */
    int   culling;           /* 0 = conservative candidate culling before the exact tests (default; results identical),
                                1 = brute force over all primitives (per-bounce kernels) */
    int   batch;             /* iterations that may share one launch group in pt_render (1 = one iteration per launch; 0 = auto:
                                about 96 M rays per launch for the whole-path kernels, 32 M for the per-bounce kernels, whose two
                                ray pools grow with it; at most 128 iterations).  Results are identical: each in-flight iteration
                                accumulates into its own plane of the owned pixels (12 bytes each: up to 1.15 GB per stream at
                                1080p with the auto group, never more than 4 GiB or a quarter of the free device memory) and the
                                planes are folded into the image in iteration order. */
    int   ordering;          /* which kernel family renders; results are identical.
                                2 = whole paths, ONE launch per group of iterations (pt_config_default): waves draw jobs of camera
                                    rays and keep every ray from the camera to its end; up to 32 primitives on typed work queues
                                    (k_path_q, also with registered meshes or direct_light), more than 32 analytic primitives --
                                    any number -- on a uniform grid with dense (ray, cell) and (ray, primitive) pairs (k_path_w:
                                    byte ids and the geometry table in LDS up to 256 primitives, wide ids and gathers from global
                                    memory beyond).
                                1 = typed work queues, one launch per bounce (<= 32 primitives).
                                0 = stable: one launch per bounce, the compacted ray stream keeps generation order -- what the
                                    parity hooks (pt_debug_trace_pool, pt_debug_primary_hits) compare against.
                                Scenes a family does not take fall back to the next one down.  Other values behave like 0. */
    int   direct_light;      /* 1 = next-event estimation: at every diffuse hit one shadow
                                ray to a point drawn by getRandomPointOnCube / getRandomPointOnSphere
                                (src/intersections.h:220-286) on a random emitter; emitter hits then add
                                radiance only for camera rays and after specular events.  Same expectation
                                as mode 0 without it, far less noise for small lights.  Needs culling = 0. */
    int   streams;           /* n > 1 (pt_config_default: 2): the context shards its rows once more over n internal contexts,
                                each on its own stream, all rendering into the same image and all enqueued before any is
                                awaited: the tails of one stream's launches are filled by the others' (mode 0; bit-identical).
                                1 = one HIP stream; the parity hooks need 1.  DESIGN.md section 4. */
    int   grid_density;      /* k_path_w: cells of its uniform grid per small primitive (0 = default 4; 1..64).  The grid only
                                decides which bounds a ray tests; results identical. */
} pt_config;

typedef struct pt_context pt_context;

/* per-kernel timing collected when cfg.profile = 1 (HIP events on the render stream) */
typedef struct {
    double generate_ms;      /* sums over the launches since pt_reset_stats */
    double bounce_ms;        /* all trace+scatter+compact launches */
    double display_ms;
    uint64_t generate_launches;
    uint64_t bounce_launches;
    uint64_t display_launches;
    uint64_t iterations;
    uint64_t live[65];       /* live[k] = rays entering bounce k, summed over iterations;
                                live[max_depth] = paths alive when the depth ran out */
    uint64_t emitted;        /* paths that ended on an emitter (accumulator read+write) */
} pt_stats;

int         pt_abi_version(void);
const char *pt_last_error(void);
void        pt_config_default(pt_config *cfg);
int         pt_device_count(void);               /* HIP devices visible (0 without a GPU) */

int  pt_create(const pt_config *cfg, pt_context **out);
void pt_destroy(pt_context *ctx);

/* Triangle data of a MESH primitive (GEOMTYPE 2, src/sceneStructs.h:14).  The reference tags `*.obj` objects in its
 * parser (src/scene.cpp:55-64) but neither loads the file nor intersects the type (src/raytraceKernel.cu:144-145 is
 * empty) and `geom` has no field for the data, so this has no counterpart in cudaRaytraceCore's arguments: callers
 * with meshes (the library's own loader, ptrender) register them here.  Object space; indices 0-based. */
typedef struct {
    int          geom_index;       /* index into the pt_geom array of pt_upload_scene; that geom must have type 2 */
    const float *vertices;         /* nvertices x 3 */
    int          nvertices;
    const int   *indices;          /* ntriangles x 3 */
    int          ntriangles;
} pt_mesh;
/* Copies the meshes; they take effect at the next pt_upload_scene (which builds one BVH per mesh).  nmeshes = 0
 * clears.  A type-2 geom without a registered mesh is skipped, like every MESH in the reference.  direct_light with
 * an emitting mesh is refused. */
int  pt_set_meshes(pt_context *ctx, const pt_mesh *meshes, int nmeshes);

/* Scene for one frame == the packing cudaRaytraceCore does at src/raytraceKernel.cu:179-206.
 * Copies; the caller keeps ownership.  (Re)allocates the ray pool for the resolution. */
int  pt_upload_scene(pt_context *ctx, const pt_geom *geoms, int ngeoms,
                     const pt_material *materials, int nmaterials, const pt_camera *camera);

/* Accumulator = camera::image (glm::vec3[W*H], index x+y*W, y=0 top; src/raytraceKernel.cu:176).
 * host_rgb == NULL zeroes it.  */
int  pt_set_image(pt_context *ctx, const float *host_rgb);
/* Render into a caller-owned DEVICE buffer of W*H*3 floats instead (e.g. a torch tensor that is
 * later reduced with RCCL); NULL returns to the internal buffer. */
int  pt_bind_device_image(pt_context *ctx, void *device_rgb);
int  pt_get_image(pt_context *ctx, float *host_rgb);       /* synchronises; the SUM over iterations */
/* Row-sharded renders (row_offset/row_stride): copy ONLY the rows this context owns into the full-frame
 * host image (W*H*3 floats; the other rows are left untouched), i.e. 1/row_stride of the frame over PCIe.
 * Calling it on every context of an N-GPU render assembles camera::image (src/raytraceKernel.cu:215 downloads
 * the whole frame from its one GPU).  Synchronises. */
int  pt_get_rows(pt_context *ctx, float *host_rgb);
/* The same exchange device to device: the rows `src` owns are copied into `dst`'s accumulator (peer copy over
 * xGMI when the contexts sit on different GPUs; both contexts must hold the same resolution).  Afterwards
 * `dst` can show or download the assembled frame (pt_display / pt_get_image).  Synchronises both. */
int  pt_gather_rows_peer(pt_context *dst, pt_context *src);

/* Enqueue iterations [first, first+count) (1-based like main.cpp:110) on the context's stream;
 * asynchronous.  Each adds one path per owned pixel into the accumulator. */
int  pt_render(pt_context *ctx, int first_iteration, int count);
int  pt_sync(pt_context *ctx);

/* sendImageToPBO (src/raytraceKernel.cu:88-119): (accumulator*scale)*255, clamp above only,
 * truncate, {x=r,y=g,z=b,w=0}.  scale = 1 is the reference.  out may be host or device memory
 * (W*H*4 bytes); NULL is ignored.  Synchronises when out is host memory. */
int  pt_display(pt_context *ctx, float scale, void *out_xyzw, int out_is_device);

/* switch the per-launch HIP-event bracketing on or off (same as cfg.profile) */
int  pt_set_profiling(pt_context *ctx, int enabled);
int  pt_get_stats(pt_context *ctx, pt_stats *out);          /* synchronises */
int  pt_reset_stats(pt_context *ctx);
int  pt_get_resolution(pt_context *ctx, int *w, int *h, int *owned_pixels);

/* ---- parity hooks (used by tests; cheap, not on the render path) ------------------- */

/* Primary ray + nearest hit for every pixel of the frame (the body of raytraceRay without the
 * colour write): dir[W*H*3], hit[W*H] (-1 = miss), t[W*H], P[W*H*3], N[W*H*3]; any may be NULL. */
int  pt_debug_primary_hits(pt_context *ctx, float *dir, int *hit, float *t, float *P, float *N);
/* Ray pool after `bounces` bounces of one iteration, compacted in generation order.  Arrays of
 * capacity owned_pixels (any may be NULL); *count receives the number of live rays. */
int  pt_debug_trace_pool(pt_context *ctx, int iteration, int bounces, int *count,
                         float *ox, float *oy, float *oz, float *dx, float *dy, float *dz,
                         float *tr, float *tg, float *tb, uint32_t *pixel);
/* The whole-path kernels (ordering = 2) bound the scheduling turns a wave may take, so that a broken build ends with
 * PT_ERR_HIP ("turn limit reached") at the next pt_sync instead of hanging the device.  0 = sized by the launch (default);
 * tests lower it to provoke the guard. */
int  pt_debug_set_turn_limit(pt_context *ctx, unsigned int turns);
/* How the uploaded scene runs on the whole-path kernels (ordering = 2): out[0] = kernel family (0 none: per-bounce kernels,
 * 1 k_path_q, 2 k_path_w), out[1] = waves per block, out[2] = blocks per CU, out[3] = LDS bytes per block, out[4] = queue
 * records (k_path_q) or ray slots (k_path_w) per wave, out[5] = arena bytes, out[6] = 1 with meshes, out[7] = 1 with direct
 * light.  A context with streams > 1 reports its first internal context. */
int  pt_debug_path_shape(pt_context *ctx, unsigned int *out8);
/* generateRandomNumberFromThread (src/raytraceKernel.cu:30-37) evaluated on the device for n
 * (x,y) pairs. */
int  pt_debug_rng_from_thread(pt_context *ctx, float resx, float resy, float time, int n,
                              const int *xy, float *out3);
/* device evaluations of the scatter primitives for n inputs (interactions.h signatures) */
int  pt_debug_hemisphere(pt_context *ctx, int n, const float *normal3, const float *xi2, float *out3);
int  pt_debug_sincos(pt_context *ctx, int n, const float *a, float *s, float *c);
/* getRandomPointOnCube / getRandomPointOnSphere (src/intersections.h:220-286) on primitive `geom`
 * of the uploaded scene for n float seeds (the reference's light-sampling helpers; no call sites there) */
int  pt_debug_light_points(pt_context *ctx, int geom, int n, const float *seeds, float *out3);

/* The spatial index of the whole-path kernel for more than 32 primitives (k_path_w), probed WITHOUT a device: builds the uniform
 * grid of the scene as pt_upload_scene does (16-bit references up to 256 primitives, 32-bit ones beyond) and walks nrays rays (6
 * floats each: origin, direction) on the host with the kernel's own walk functions.  out_sets: nrays x max(8, ceil(ngeoms / 32))
 * words, bit p of a ray = primitive p gets its bound tested for that ray (every
 * bit for a ray the kernel would not walk: it takes the reference loop).  out_info: [0] cells [1] references [2] big
 * primitives [3] primitives listed twice for a ray (must be 0) [4] rays not walked [5..7] cells per axis [8] mean and
 * [9] longest walk in cells [10] non-empty cells and [11] listed primitives per ray x 100 [12] bytes of LDS [13] [14] the walk
 * lengths that separate the three bins survivors are sorted into [15] worst and [16] mean x 100 error of the length estimate
 * behind that sorting (cells).  out_info holds 24 words. */
int  pt_debug_grid_probe(const pt_geom *geoms, int ngeoms, int density, const float *rays, int nrays,
                         uint32_t *out_sets, uint32_t *out_info);
/* The cone test k_path_w gives a group of 64 camera rays instead of 64 grid walks, on the host: nfans x 64 rays (6 floats each;
 * a fan = 64 rays with a common origin).  out_sets: nfans x max(8, ceil(ngeoms / 32)) words, bit p = the cone of the fan meets
 * primitive p's bound, i.e. the fan's rays test that bound (every bit when the fan gets no cone -- no common origin, a half angle above
 * 5.7 degrees, not finite: the kernel then walks the grid).  out_info[0] = fans that got a cone; out_info holds 4 words. */
int  pt_debug_fan_probe(const pt_geom *geoms, int ngeoms, const float *rays, int nfans, uint32_t *out_sets, uint32_t *out_info);

/* ---- host-side scene I/O (no GPU needed; src/scene.cpp grammar, src/image.cpp output) ---- */

typedef struct pt_scene pt_scene;

int  pt_scene_load(const char *path, pt_scene **out);       /* PT_ERR_IO / PT_ERR_PARSE */
void pt_scene_free(pt_scene *s);
int  pt_scene_counts(const pt_scene *s, int *ngeoms, int *nmaterials, int *nframes, int *iterations);
const char *pt_scene_image_name(const pt_scene *s);
/* flatten frame `frame` the way cudaRaytraceCore does; arrays sized by pt_scene_counts */
int  pt_scene_flatten(const pt_scene *s, int frame, pt_geom *geoms, pt_material *materials, pt_camera *camera);
/* MESH objects of the file: the `<name>.obj` named on the object's type line is read relative to the scene file
 * (`v x y z` and `f a b c ...` lines; polygons are fanned; a/b/c index forms and negative indices accepted).
 * pt_scene_mesh fills *mesh with pointers owned by the scene (valid until pt_scene_free). */
int  pt_scene_mesh_count(const pt_scene *s);
int  pt_scene_mesh(const pt_scene *s, int k, pt_mesh *mesh);
/* full 4x4 rows of transform / inverse of object i at frame f (parity with scene.cpp:123-125) */
int  pt_scene_object_matrices(const pt_scene *s, int object, int frame, float transform16[16], float inverse16[16]);
/* buildTransformationMatrix + inverse (src/utilities.cpp:70-86) */
int  pt_build_transform(const float t[3], const float r[3], const float s[3], float transform16[16], float inverse16[16]);

/* gamma + clamp + u8 exactly as image::saveImageRGB (src/image.cpp:40-87) with the settings of
 * main.cpp:143-147; out_rgb holds W*H*3 bytes, top row first. */
int  pt_image_to_u8(const float *rgb_sum, int w, int h, int divisor, float gamma, uint8_t *out_rgb);
/* writes .bmp (24-bit, bottom-up, like stbi_write_bmp) or .png by extension */
int  pt_image_save(const char *path, const float *rgb_sum, int w, int h, int divisor, float gamma);

#ifdef __cplusplus
}
#endif
#endif /* PTMI355_H */
