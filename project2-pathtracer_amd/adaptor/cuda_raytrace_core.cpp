// cuda_raytrace_core.cpp -- the ONE symbol the reference's host code calls on the hot path,
//
//     void cudaRaytraceCore(uchar4* pos, camera* renderCam, int frame, int iterations,
//                           material* materials, int numberOfMaterials,
//                           geom* geoms, int numberOfGeoms);
//
// (declared /root/reference/src/raytraceKernel.h:18, defined src/raytraceKernel.cu:164-227, only
// caller src/main.cpp:126), re-implemented on top of the C ABI of libptmi355.so.
//
// Build this TU with the SAME compiler, C++ runtime and headers as the reference's main.cpp and
// scene.cpp: it includes the reference's own sceneStructs.h (so `camera`, which embeds a
// std::string, `geom`, `material` and `uchar4` are the caller's types and the mangled name
// matches), and it crosses into the HIP library only through include/ptmi355.h (PODs, no HIP
// headers here).  See INTEGRATION.md.
//
// Semantics kept from the reference wrapper:
//   * `iterations` is the 1-based index of this call (main.cpp:110) and seeds the sample;
//   * renderCam->image is in/out: it is uploaded when a frame starts (iterations == 1: main.cpp
//     zeroes it between frames, :168-170) and holds the SUM over iterations when main.cpp reads
//     it (:136-147) -- downloaded on the final iteration, or every call with PT_SYNC_EVERY_CALL=1;
//   * `pos` receives sendImageToPBO's bytes (raytraceKernel.cu:88-119) when non-NULL;
//   * errors print "Cuda error: <what>: <why>." and exit(EXIT_FAILURE) (raytraceKernel.cu:20-26).
// Options the reference has no channel for come from the environment:
//   PT_MODE=pathtrace|reference  PT_MAX_DEPTH  PT_CAMERA_MODE  PT_AA  PT_APERTURE  PT_FOCAL_DIST
//   PT_DIRECT_LIGHT  PT_STREAMS  PT_ORDERING  PT_DEVICE  PT_PBO_IS_DEVICE  PT_SYNC_EVERY_CALL  PT_LAZY_BATCH  PT_NGPU  PT_DEVICES
//   PT_DUMP_IMAGE=<file> (test hook: camera::image as raw floats after the final iteration)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sceneStructs.h"   // the reference's header (-I<reference>/src)
#include "ptmi355.h"

static_assert(sizeof(material) == sizeof(pt_material), "material layout differs from pt_material");
static_assert(sizeof(cudaMat4) == 64, "cudaMat4 must be four rows of four floats");

namespace {

// One context per GPU (PT_NGPU, default 1; PT_DEVICES="0,1,..." picks the ordinals): the frame's rows
// are interleaved over them (row y -> context y % N), every pt_render only ENQUEUES work on its own
// device's stream, so the GPUs run concurrently from this single caller thread; the owned rows are
// gathered into camera::image when the caller can observe it.  (bench.py uses one process per GPU and an
// RCCL reduce instead; inside the reference's single-process viewer this is the equivalent.)
std::vector<pt_context *> g_ctxs;
#define g_ctx (g_ctxs[0])
unsigned long long g_scene_hash = 0;
int g_mode = 0;
// Lazy batching behind the per-iteration API (PT_LAZY_BATCH=K, headless runs): calls only queue their
// iteration index; the queued run is rendered as ONE pt_render(first, count) -- which the library
// executes as a single launch group -- when K calls have accumulated, when the caller can observe the
// result (pos != NULL, final iteration, PT_SYNC_EVERY_CALL) or when the scene changes.  What main.cpp
// can see (camera::image after the last iteration, src/main.cpp:136-147) is unchanged.
int g_pending_first = 0, g_pending_count = 0;
// options read from the environment ONCE, when the contexts are created (not per call)
int g_sync_every_call = 0, g_lazy = 1, g_pbo_is_device = 0;

void flush_pending() {
    if (g_pending_count > 0) {
        for (pt_context *c : g_ctxs)
            if (pt_render(c, g_pending_first, g_pending_count) != PT_OK) {
                fprintf(stderr, "Cuda error: %s: %s.\n", "pt_render", pt_last_error());
                exit(EXIT_FAILURE);
            }
        g_pending_count = 0;
    }
}

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "Cuda error: %s: %s.\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

float env_float(const char *name, float dflt) {
    const char *v = getenv(name);
    return (v && *v) ? (float)atof(v) : dflt;
}

unsigned long long fnv1a(const void *p, size_t n, unsigned long long h) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

void rows3(const cudaMat4 &m, float out[12]) {
    const glm::vec4 r[3] = {m.x, m.y, m.z};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = r[i][j];
}

}  // namespace

// the shim's cudaDeviceReset() (main.cpp:159,171) should call this: drops all device state
extern "C" void ptmi355_adaptor_reset(void) {
    g_pending_count = 0;
    for (pt_context *c : g_ctxs) pt_destroy(c);
    g_ctxs.clear();
    g_scene_hash = 0;
}

namespace {

// camera::image <- the rows each context owns (every context started from the same host image, so a
// row's owner holds initial value + all of that row's samples)
void gather_image(float *host, int W, int H) {
    (void)W; (void)H;
    if (g_ctxs.size() == 1) {
        if (pt_get_image(g_ctxs[0], host) != PT_OK) die("Kernel failed!");
        return;
    }
    // N GPUs: each sends only its own rows (1/N of the frame over its own PCIe link)
    for (pt_context *c : g_ctxs)
        if (pt_get_rows(c, host) != PT_OK) die("Kernel failed!");
}

}  // namespace

void cudaRaytraceCore(uchar4 *PBOpos, camera *renderCam, int frame, int iterations, material *materials,
                      int numberOfMaterials, geom *geoms, int numberOfGeoms) {
    if (g_ctxs.empty()) {
        const char *mode = getenv("PT_MODE");
        g_mode = (mode && !strcmp(mode, "reference")) ? 1 : 0;
        int ngpu = env_int("PT_NGPU", 1);
        if (ngpu < 1) ngpu = 1;
        std::vector<int> devices;
        if (const char *list = getenv("PT_DEVICES")) {
            for (const char *p = list; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
        }
        for (int r = 0; r < ngpu; ++r) {
            pt_config cfg;
            pt_config_default(&cfg);
            cfg.mode = g_mode;
            cfg.device = r < (int)devices.size() ? devices[r] : env_int("PT_DEVICE", 0) + r;
            cfg.max_depth = env_int("PT_MAX_DEPTH", 8);
            cfg.camera_mode = env_int("PT_CAMERA_MODE", 0);
            cfg.antialias = env_int("PT_AA", 0);
            cfg.aperture = env_float("PT_APERTURE", 0.0f);
            cfg.focal_distance = env_float("PT_FOCAL_DIST", 0.0f);
            cfg.direct_light = env_int("PT_DIRECT_LIGHT", 0);
            cfg.streams = env_int("PT_STREAMS", 2);
            cfg.ordering = env_int("PT_ORDERING", 2);         // whole paths on the typed work queues: fastest, results identical
            cfg.row_offset = r;
            cfg.row_stride = ngpu;
            pt_context *c = nullptr;
            if (pt_create(&cfg, &c) != PT_OK) die("pt_create");
            g_ctxs.push_back(c);
        }
        g_sync_every_call = env_int("PT_SYNC_EVERY_CALL", 0);
        g_lazy = env_int("PT_LAZY_BATCH", 1);
        g_pbo_is_device = env_int("PT_PBO_IS_DEVICE", 0);
    }

    // pack the frame exactly as the reference wrapper does (raytraceKernel.cu:179-206)
    std::vector<pt_geom> pg(numberOfGeoms);
    for (int i = 0; i < numberOfGeoms; ++i) {
        pg[i].type = (int)geoms[i].type;
        pg[i].materialid = geoms[i].materialid;
        rows3(geoms[i].transforms[frame], pg[i].transform);
        rows3(geoms[i].inverseTransforms[frame], pg[i].inverseTransform);
    }
    pt_camera cam;
    cam.resolution[0] = renderCam->resolution.x; cam.resolution[1] = renderCam->resolution.y;
    for (int k = 0; k < 3; ++k) {
        cam.position[k] = renderCam->positions[frame][k];
        cam.view[k] = renderCam->views[frame][k];
        cam.up[k] = renderCam->ups[frame][k];
    }
    cam.fov[0] = renderCam->fov.x; cam.fov[1] = renderCam->fov.y;

    // the caller re-allocates geoms/materials every call (main.cpp:114-122): key the device copy
    // on CONTENT, never on the pointers
    unsigned long long h = fnv1a(pg.data(), pg.size() * sizeof(pt_geom), 1469598103934665603ull);
    h = fnv1a(materials, (size_t)numberOfMaterials * sizeof(material), h);
    h = fnv1a(&cam, sizeof cam, h);
    const int W = (int)renderCam->resolution.x, H = (int)renderCam->resolution.y;
    if (h != g_scene_hash) {
        flush_pending();
        // Geometry, materials or camera changed in the middle of a frame: the reference round-trips camera::image on
        // every call (raytraceKernel.cu:176,215), so the samples rendered so far live on in the host image.  Here they
        // live on the device until an observation point: bring them home before the device state is rebuilt from it.
        if (g_scene_hash != 0 && iterations > 1 && g_mode == 0) {
            int w0 = 0, h0 = 0, own = 0;
            if (pt_get_resolution(g_ctx, &w0, &h0, &own) == PT_OK && w0 == W && h0 == H)
                gather_image(reinterpret_cast<float *>(renderCam->image), W, H);
        }
        for (pt_context *c : g_ctxs) {
            if (pt_upload_scene(c, pg.data(), numberOfGeoms, reinterpret_cast<const pt_material *>(materials),
                                numberOfMaterials, &cam) != PT_OK) die("pt_upload_scene");
            if (pt_set_image(c, reinterpret_cast<const float *>(renderCam->image)) != PT_OK) die("pt_set_image");
        }
        g_scene_hash = h;
    } else if (iterations == 1) {
        for (pt_context *c : g_ctxs)
            if (pt_set_image(c, reinterpret_cast<const float *>(renderCam->image)) != PT_OK) die("pt_set_image");
    }

    const bool final_call = (unsigned)iterations >= renderCam->iterations;
    const bool observable = PBOpos != NULL || final_call || g_mode == 1 || g_sync_every_call;
    const int lazy = g_lazy;
    if (g_pending_count > 0 && g_pending_first + g_pending_count != iterations) flush_pending();   // not consecutive
    if (g_pending_count == 0) g_pending_first = iterations;
    g_pending_count++;
    if (observable || g_pending_count >= lazy) flush_pending();

    const bool download = final_call || g_mode == 1 || g_sync_every_call;
    if (PBOpos) {
        const float scale = g_mode == 1 ? 1.0f : 1.0f / (float)iterations;
        if (g_ctxs.size() == 1) {
            if (pt_display(g_ctx, scale, PBOpos, g_pbo_is_device) != PT_OK) die("pt_display");
        } else {
            // several GPUs: the owned rows travel device to device (peer copies over xGMI) into context 0's
            // accumulator, which then runs sendImageToPBO like the single-GPU case
            for (size_t r = 1; r < g_ctxs.size(); ++r)
                if (pt_gather_rows_peer(g_ctx, g_ctxs[r]) != PT_OK) die("pt_gather_rows_peer");
            if (pt_display(g_ctx, scale, PBOpos, g_pbo_is_device) != PT_OK) die("pt_display");
        }
    }
    if (download) {
        gather_image(reinterpret_cast<float *>(renderCam->image), W, H);
        // test hook: the raw accumulator the caller now holds (the viewer only ever saves its gamma-corrected bytes)
        if (final_call)
            if (const char *path = getenv("PT_DUMP_IMAGE"))
                if (FILE *f = fopen(path, "wb")) { fwrite(renderCam->image, sizeof(float) * 3, (size_t)W * H, f); fclose(f); }
    }
}
