// pt_api.hip -- host side of libptmi355.so: contexts, scene upload (tables, culling bounds, clusters, mesh BVHs),
// launch planning and the C ABI of include/ptmi355.h.
//
// Replaces the body of cudaRaytraceCore (/root/reference/src/raytraceKernel.cu:164-227); the kernels it launches
// (raytraceRay :123-159, sendImageToPBO :88-119) live in the pt_k_*.hip translation units (map: pt_kernels.hpp).
// No CPU fallback lives here: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../include/ptmi355.h"
#include "pt_kernels.hpp"
#include "pt_host.hpp"
#include "pt_build.hpp"

using namespace ptd;
using namespace ptk;

// =========================================================================== host side ==

struct pt_context {
    pt_config cfg;
    hipStream_t stream = nullptr;
    int n_cu = 0;
    int W = 0, H = 0;
    uint32_t n_own = 0, cap = 0;
    int G = 0, M = 0;
    CamRec cam;
    float *pool[2] = {nullptr, nullptr};
    float *image_own = nullptr;
    float *image = nullptr;          // bound or own
    GeomRec *d_geoms = nullptr;
    MatRec *d_mats = nullptr;
    SyncBlock *d_sync = nullptr;
    bool cull = true;                // AABB candidate culling in front of the exact tests (cfg.culling == 0)
    bool queue = false;              // typed work-queue kernel (cfg.ordering == 1 or 2; LDS geometry, G <= 32)
    bool pathq = false;              //   cfg.ordering == 2: whole paths in one launch (k_path_q), per-wave stacks instead of pools
    bool pathq_nee = false;          //   ... with direct_light (k_path_q<NEE>: shadow rays are records of the same queues); the parity hooks keep k_bounce_seg<NEE>
    bool pathw = false;              // cfg.ordering == 2 with 33..256 analytic primitives: whole paths, dense pairs (k_path_w)
    float *d_arena = nullptr;        //   [grid_path * kWaves][kSFields][kStack]: the waves' ray stacks
    uint32_t *d_tickets = nullptr;   //   [kTicketCtrs][kTicketStride]
    size_t arena_bytes = 0;
    int grid_path = 0;
    uint32_t path_static_eighths = 4; //   eighths of the camera-ray jobs every wave owns statically (half: all-static 9 % slower, all-drawn 2 %)
    uint32_t path_waves = 4;         //   waves per block of the whole-path kernel in use
    int wide_shape = 0;              //   k_path_w: block shape (0 / 1 narrow ids, 2 / 3 wide ids: pt_k_wide.hip)
    bool wide_big = false;           //   more than 256 primitives: wide ids, geometry gathered from global memory
    uint32_t wide_stack = 0, wide_slots = 0;
    float wide_qscale = 1.0f, wide_slack = 0.0f;
    GridArgs grid;                   //   k_path_w: the uniform grid over the small primitives (build_grid)
    unsigned char *d_grid = nullptr; //   its blob: cells | refs | big list
    float *d_tap = nullptr;          // parity hook of the whole-path kernels (pt_debug_trace_pool): where the rays entering bounce tap_level go
    uint32_t *d_tap_count = nullptr;
    uint32_t tap_level = 0, tap_cap = 0;
    uint32_t turn_limit = 0;         // whole-path kernels: explicit guard against a wave that never finishes (0 = by launch size; pt_debug_set_turn_limit)
    int occ_bounce = 2;              // resident blocks per CU of the per-bounce kernel in use
    uint32_t lds_path = 0;
    int path_cap = 144;              //   records per wave of k_path_q (one of kPCaps)
    bool queue_mesh = false;         //   its variant with mesh traversal (scene has MESH primitives with triangles)
    FaceFrame *d_frames = nullptr;   // [G][3] shading frames of the box primitives (k_bounce_q)
    CullRec *d_cull = nullptr;       // bounds for its culling pass, cubes first
    int q_nbox = 0, q_nsph = 0;
    // MESH primitives (pt_set_meshes): host copies, and one device blob [nodes | triangles] per mesh of the uploaded scene
    std::vector<pth::HostMesh> meshes;
    std::vector<void *> d_mesh_blobs;
    uint32_t nseg = 0, seg_slots = 0;     // capacity: segments of the smallest size in use
    uint32_t cur_slots = 0, cur_nseg = 0; // segment layout of the launch group being enqueued
    uint32_t *d_segcnt[2] = {nullptr, nullptr};
    uchar4 *d_display = nullptr;
    uint32_t lds_bytes = 0;
    int grid_bounce = 0;
    bool geom_lds = true;
    bool scene_ready = false;
    bool counts_pending = false;
    uint32_t bank = 0;               // counter bank of the iteration being enqueued (fused segmented path)
    uint32_t batch_max = 1;          // iterations that may share one launch group
    uint32_t pix_mask = 0xFFFFFFu;   // pixel bits of the pool's pixel word (all 32 for frames above 2^24 pixels)
    bool empty = false;              // this context owns no row of the frame (row_offset >= H): every call is a no-op
    float *d_planes = nullptr;       // batch_max accumulator planes (owned pixels x 3 floats each)
    bool wide = false;               // 33..256 primitives in LDS: k_bounce_seg<.., WIDE> (two-level cluster culling -> packed candidate lists)
    int nbc = 0, nsc = 0;            // its cube / sphere clusters, stored behind the GeomRec array of d_geoms
    uint32_t cluster_bytes = 0;
    // cfg.streams > 1: this context only owns the frame (image) and fans every call out to `subs`, one
    // ordinary context per stream, each rendering every streams-th of this context's rows into that image
    std::vector<pt_context *> subs;
    bool nee = false;                // cfg.direct_light: shadow rays at diffuse hits (k_bounce_seg<.., NEE>)
    uint32_t *d_lights = nullptr;    // indices of the emitting primitives
    uint32_t nlights = 0;
    // profiling
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> free_events;
    double ms[3] = {0, 0, 0};
    uint64_t launches[3] = {0, 0, 0};
    uint64_t iterations = 0;
};

namespace {

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            pth::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PT_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

hipEvent_t take_event(pt_context *c) {
    if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;    // the caller then leaves this launch untimed
    return e;
}

struct Scoped {
    pt_context *c; int kind; hipEvent_t a = nullptr, b = nullptr;
    Scoped(pt_context *ctx, int k) : c(ctx), kind(k) {
        if (c->cfg.profile) {
            a = take_event(c); b = take_event(c);
            if (a && b) (void)hipEventRecord(a, c->stream);
            else { if (a) c->free_events.push_back(a); if (b) c->free_events.push_back(b); a = b = nullptr; }
        }
    }
    ~Scoped() {
        c->launches[kind]++;
        if (a && b) { (void)hipEventRecord(b, c->stream); c->pending.push_back({a, b, kind}); }
    }
};

int resolve_events(pt_context *c) {
    for (auto &e : c->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        c->ms[e.kind] += ms;
        c->free_events.push_back(e.a);
        c->free_events.push_back(e.b);
    }
    c->pending.clear();
    return PT_OK;
}

void free_scene_buffers(pt_context *c) {
    for (int i = 0; i < 2; ++i) { if (c->pool[i]) (void)hipFree(c->pool[i]); c->pool[i] = nullptr; }
    if (c->image_own) (void)hipFree(c->image_own);
    if (c->d_geoms) (void)hipFree(c->d_geoms);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_display) (void)hipFree(c->d_display);
    for (int i = 0; i < 2; ++i) { if (c->d_segcnt[i]) (void)hipFree(c->d_segcnt[i]); c->d_segcnt[i] = nullptr; }
    if (c->d_planes) (void)hipFree(c->d_planes);
    c->d_planes = nullptr;
    if (c->d_lights) (void)hipFree(c->d_lights);
    c->d_lights = nullptr;
    if (c->d_frames) (void)hipFree(c->d_frames);
    c->d_frames = nullptr;
    if (c->d_cull) (void)hipFree(c->d_cull);
    c->d_cull = nullptr;
    if (c->d_arena) (void)hipFree(c->d_arena);
    c->d_arena = nullptr;
    if (c->d_grid) (void)hipFree(c->d_grid);
    c->d_grid = nullptr;
    if (c->d_tickets) (void)hipFree(c->d_tickets);
    c->d_tickets = nullptr;
    for (void *b : c->d_mesh_blobs) (void)hipFree(b);
    c->d_mesh_blobs.clear();
    if (c->image == c->image_own) c->image = nullptr;
    c->image_own = nullptr; c->d_geoms = nullptr; c->d_mats = nullptr; c->d_display = nullptr;
    c->scene_ready = false;
}

// Every clear of device memory goes through hipMemsetAsync on the context's stream: that stream is
// created non-blocking, so a hipMemset on the null stream would NOT be ordered against the kernels
// launched here (it once wiped the primary-hit hook's output after the kernel had written it).
// (The scene builders -- culling bounds, k_path_w's grid, clusters, mesh BVHs -- are host-only code: pt_build.cpp.)

int launch_seg(pt_context *c, const SegArgs &a, bool last, bool gen) {
    Scoped s(c, 1);
    if (c->queue) {
        QTables qt;
        qt.frames = c->d_frames; qt.cull = c->d_cull; qt.nbox = c->q_nbox; qt.nsph = c->q_nsph;
        queue_launch(c->queue_mesh, last, gen, c->grid_bounce, c->lds_bytes, c->stream, a, c->d_geoms, c->d_mats, qt);
    } else {
        const SegVariant v = {c->geom_lds, c->cull, c->nee, c->wide};
        seg_launch(v, last, gen, c->grid_bounce, c->lds_bytes, c->stream, a, c->d_geoms, c->d_mats);
    }
    HIPCHK(hipGetLastError());
    return PT_OK;
}

// Segment size for a launch group: fixed by cfg.chunk_rays, else about four segments per resident
// wave, a multiple of 64 (full wave groups) between 192 and 1024 -- small launches need many small
// segments to occupy every wave, big (batched) launches run best on long ones (measured).
uint32_t seg_slots_for(const pt_context *c, uint32_t n_rays) {
    if (c->cfg.chunk_rays > 0) return c->seg_slots;
    // the typed-queue kernel drains once per segment (one partly filled group): longer segments there
    const bool longseg = c->queue;
    const uint32_t slots = (uint32_t)c->grid_bounce * kWaves * (longseg ? 2u : 4u);
    uint32_t S = (((n_rays + slots - 1) / slots) + 63u) & ~63u;
    if (S < 192u) S = 192u;
    if (S > (longseg ? 2048u : 1024u)) S = longseg ? 2048u : 1024u;
    return S;
}

int enqueue_fold(pt_context *c, uint32_t batch) {
    Scoped s(c, 0);
    FoldArgs f;
    f.image = c->image; f.planes = c->d_planes; f.plane_stride = (size_t)c->n_own * 3;
    f.batch = batch; f.n_own = c->n_own; f.W = c->W; f.row_offset = c->cfg.row_offset; f.row_stride = c->cfg.row_stride;
    fold_launch(c->stream, f);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

// `batch` consecutive iterations starting at `iteration` as ONE launch group.  stop_after < 0 renders all
// bounces, otherwise only the first `stop_after` bounces without the LAST variant (parity hook, batch = 1,
// always on the per-bounce kernels).
int enqueue_iterations(pt_context *c, uint32_t iteration, uint32_t batch, int stop_after) {
    const int D = c->cfg.max_depth;
    // Camera rays are generated inside the first bounce launch; k_generate runs only for the parity hook
    // that wants the pool before any bounce.
    const bool fused = stop_after != 0;
    const uint32_t n_rays = batch * c->n_own;
    c->cur_slots = seg_slots_for(c, n_rays);
    c->cur_nseg = (n_rays + c->cur_slots - 1) / c->cur_slots;
    if (!fused) {
        Scoped s(c, 0);
        GenArgs g;
        g.cam = c->cam; g.pool = c->pool[0]; g.cap = c->cap; g.n_own = c->n_own; g.iteration = iteration;
        g.sync = c->d_sync;
        g.seg_cnt0 = c->d_segcnt[0]; g.nseg = c->cur_nseg; g.seg_slots = c->cur_slots;
        generate_launch(c->stream, g, c->n_own > g.nseg ? c->n_own : g.nseg);
        HIPCHK(hipGetLastError());
    }
    const int nb = stop_after < 0 ? D : stop_after;
    if (fused) c->bank ^= 1u;
    if ((c->pathq || c->pathw) && stop_after < 0) {
        // whole paths: one launch for the group (the per-bounce launches below remain the parity hooks' path)
        SegArgs a;
        memset(&a, 0, sizeof a);
        a.cap = c->cap; a.image = c->image; a.G = c->G; a.M = c->M; a.sync = c->d_sync;
        a.iteration = iteration; a.n_own = c->n_own; a.cam = c->cam; a.bank = c->bank; a.pix_mask = c->pix_mask;
        a.n_rays = n_rays; a.batch = batch; a.planes = c->d_planes; a.plane_stride = (size_t)c->n_own * 3;
        a.nbc = c->nbc; a.nsc = c->nsc; a.cluster_bytes = c->cluster_bytes;
        a.lights = c->d_lights; a.nlights = c->nlights;
        PathArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.arena = c->d_arena; pa.arena_bytes = c->arena_bytes < (1ull << 32) ? (uint32_t)c->arena_bytes : 0u;
        pa.depth = (uint32_t)D; pa.ticket = c->d_tickets; pa.error = &c->d_sync->error;
        pa.qscale = c->wide_qscale; pa.slack_max = c->wide_slack;
        pa.tap = c->d_tap; pa.tap_count = c->d_tap_count; pa.tap_level = c->tap_level; pa.tap_cap = c->tap_cap;
        const uint64_t waves = (uint64_t)c->grid_path * (uint64_t)c->path_waves;
        {   // job size: at least 48 jobs per wave of the launch, whole groups, 64 .. kJobMax rays (rounded DOWN: the last jobs a wave
            // draws decide how long the launch's tail is -- 64 instead of 128 rays at the driver's 20 steps: 0.1691 -> 0.1670 ms/step)
            const uint64_t per_wave = (uint64_t)n_rays / (waves * 48u);
            uint32_t job = (uint32_t)(per_wave & ~63ull);
            if (job < 64u) job = 64u;
            if (job > kJobMax) job = kJobMax;
            if (c->cfg.chunk_rays > 0) job = (uint32_t)((c->cfg.chunk_rays + 63) & ~63);      // explicit
            pa.job_rays = job;
            // static share of the jobs: half
            const uint64_t njobs = ((uint64_t)n_rays + job - 1) / job;
            pa.static_rounds = (uint32_t)(njobs * c->path_static_eighths / 8u / waves);
        }
        {   // a wave that takes far more scheduling turns than its share of the launch can need gives up with error 3 instead of
            // hanging the device.  The guard only has to bound a broken build, so it is priced at the real worst case: a live
            // ray-bounce costs one FRESH turn and at most one exact test per candidate primitive (k_path_q: G <= 32 candidates; k_path_w:
            // the 8 entries of a list, an overflowing ray one confirming test), a turn may serve a single ray, direct light doubles
            // the records of a bounce -- groups x 64 x (candidates + 2) x (1 or 2), dynamic jobs may give one wave four times its share;
            // pt_debug_set_turn_limit overrides (tests)
            const uint64_t groups = ((uint64_t)n_rays * (uint64_t)D / 64u) / waves + 1024u;
            const uint64_t cand = c->pathw ? 10u : (uint64_t)(c->G < 32 ? c->G : 32) + 2u;
            const uint64_t lim = groups * 64u * cand * (c->pathq_nee ? 2u : 1u) * 4u;
            pa.turn_limit = c->turn_limit ? c->turn_limit : (lim > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)lim);
        }
        QTables qt;
        qt.frames = c->d_frames; qt.cull = c->d_cull; qt.nbox = c->q_nbox; qt.nsph = c->q_nsph;
        HIPCHK(hipMemsetAsync(pa.ticket, 0, (size_t)kTicketCtrs * kTicketStride * sizeof(uint32_t), c->stream));
        {
            Scoped s(c, 1);
            if (c->pathw) wide_launch(c->wide_shape, c->grid_path, c->lds_path, c->stream, a, pa, c->grid, c->d_geoms, c->d_mats, c->d_frames);
            else path_launch(c->queue_mesh, c->pathq_nee, c->path_cap, c->grid_path, c->lds_path, c->stream, a, pa, c->d_geoms, c->d_mats, qt);
            HIPCHK(hipGetLastError());
        }
        if (batch > 1u || c->nee) { int rc = enqueue_fold(c, batch); if (rc) return rc; }
        c->counts_pending = true;
        return PT_OK;
    }
    for (int b = 0; b < nb; ++b) {
        SegArgs a;
        memset(&a, 0, sizeof a);
        a.in = c->pool[b & 1]; a.out = c->pool[(b + 1) & 1]; a.cap = c->cap; a.image = c->image;
        a.G = c->G; a.M = c->M; a.sync = c->d_sync;
        a.cnt_in = c->d_segcnt[b & 1]; a.cnt_out = c->d_segcnt[(b + 1) & 1];
        a.nseg_in = c->cur_nseg; a.nseg_out = c->cur_nseg; a.seg_slots = c->cur_slots;
        a.bounce = b; a.iteration = iteration; a.n_own = c->n_own; a.cam = c->cam; a.bank = c->bank;
        a.pix_mask = c->pix_mask;
        { const uint64_t pb = (uint64_t)c->cap * kFields * sizeof(float); a.pool_bytes = (c->queue && pb < (1ull << 32)) ? (uint32_t)pb : 0u; }
        a.n_rays = n_rays; a.batch = batch; a.planes = c->d_planes; a.plane_stride = (size_t)c->n_own * 3;
        a.lights = c->d_lights; a.nlights = c->nlights;
        a.nbc = c->nbc; a.nsc = c->nsc; a.cluster_bytes = c->wide ? c->cluster_bytes : 0u;
        const bool last = (stop_after < 0) && (b == D - 1);
        int rc = launch_seg(c, a, last, b == 0);
        if (rc) return rc;
    }
    if (batch > 1u || c->nee) { int rc = enqueue_fold(c, batch); if (rc) return rc; }
    c->counts_pending = true;
    return PT_OK;
}

int check_device_error(pt_context *c) {
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, &c->d_sync->error, sizeof err, hipMemcpyDeviceToHost));
    if (err == 2u || err == 3u) { pth::set_error("whole-path kernel: %s (device state corrupt)", err == 2u ? "a level ring overflowed" : "turn limit reached"); return PT_ERR_HIP; }
    if (err) { pth::set_error("a kernel reported device error %u (device state corrupt)", err); return PT_ERR_HIP; }
    return PT_OK;
}

}  // namespace

// ---- cfg.streams > 1 ------------------------------------------------------------------------------
// Row sharding inside one GPU ("Two contexts per GPU"): the sub-contexts are plain
// contexts with row_offset/row_stride refined by the stream index; they share the parent's image.
namespace multi {

int for_all(pt_context *c, int (*fn)(pt_context *)) {
    for (pt_context *s : c->subs) { int rc = fn(s); if (rc) return rc; }
    return PT_OK;
}

int rebind(pt_context *c) {
    for (pt_context *s : c->subs) { int rc = pt_bind_device_image(s, c->image); if (rc) return rc; }
    return PT_OK;
}

}  // namespace multi

extern "C" {

int pt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pt_create(const pt_config *cfg, pt_context **out) {
    if (!cfg || !out) { pth::set_error("pt_create: null argument"); return PT_ERR_ARGUMENT; }
    *out = nullptr;
    // argument checks need no device
    if (cfg->max_depth < 1 || cfg->max_depth > 64) { pth::set_error("pt_create: max_depth %d not in 1..64", cfg->max_depth); return PT_ERR_ARGUMENT; }
    if (cfg->row_stride < 1 || cfg->row_offset < 0 || cfg->row_offset >= cfg->row_stride) { pth::set_error("pt_create: bad row_offset/row_stride"); return PT_ERR_ARGUMENT; }
    if (cfg->grid_density < 0 || cfg->grid_density > 64) { pth::set_error("pt_create: grid_density %d not in 0..64", cfg->grid_density); return PT_ERR_ARGUMENT; }
    if (cfg->streams < 1 || cfg->streams > 8) { pth::set_error("pt_create: streams %d not in 1..8", cfg->streams); return PT_ERR_ARGUMENT; }
    if (cfg->batch < 0 || cfg->chunk_rays < 0 || cfg->blocks_per_cu < 0) { pth::set_error("pt_create: batch / chunk_rays / blocks_per_cu must not be negative"); return PT_ERR_ARGUMENT; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        pth::set_error("pt_create: no HIP device visible (this library has no CPU fallback)");
        return PT_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= n) { pth::set_error("pt_create: device %d out of range (%d visible)", cfg->device, n); return PT_ERR_ARGUMENT; }
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pth::set_error("pt_create: device %d is %s; this library carries gfx950 code objects only", cfg->device, prop.gcnArchName);
        return PT_ERR_NO_DEVICE;
    }
    if (cfg->streams > 1 && cfg->mode == 0) {
        pt_context *parent = new pt_context();
        parent->cfg = *cfg;
        if (hipStreamCreateWithFlags(&parent->stream, hipStreamNonBlocking) != hipSuccess) { delete parent; pth::set_error("hipStreamCreate failed"); return PT_ERR_HIP; }
        for (int r = 0; r < cfg->streams; ++r) {
            pt_config sub = *cfg;
            sub.streams = 1;
            sub.row_offset = cfg->row_offset + r * cfg->row_stride;
            sub.row_stride = cfg->row_stride * cfg->streams;
            pt_context *sc = nullptr;
            int rc = pt_create(&sub, &sc);
            if (rc) { pt_destroy(parent); return rc; }
            parent->subs.push_back(sc);
        }
        *out = parent;
        return PT_OK;
    }
    pt_context *c = new pt_context();
    c->cfg = *cfg;
    c->n_cu = prop.multiProcessorCount;
    c->geom_lds = true;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; pth::set_error("hipStreamCreate failed"); return PT_ERR_HIP; }
    if (hipMalloc(&c->d_sync, sizeof(SyncBlock)) != hipSuccess) { delete c; pth::set_error("hipMalloc(sync) failed"); return PT_ERR_HIP; }
    (void)hipMemsetAsync(c->d_sync, 0, sizeof(SyncBlock), c->stream);
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_context *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    for (pt_context *s : c->subs) pt_destroy(s);
    c->subs.clear();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene_buffers(c);
    if (c->d_sync) (void)hipFree(c->d_sync);
    for (auto &e : c->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_set_meshes(pt_context *c, const pt_mesh *meshes, int nmeshes) {
    if (!c || nmeshes < 0 || (nmeshes > 0 && !meshes)) { pth::set_error("pt_set_meshes: bad argument"); return PT_ERR_ARGUMENT; }
    std::vector<pth::HostMesh> copy;
    for (int i = 0; i < nmeshes; ++i) {
        const pt_mesh &m = meshes[i];
        if (!m.vertices || !m.indices || m.nvertices < 3 || m.ntriangles < 1 || m.ntriangles >= (1 << 27) || m.geom_index < 0) {
            pth::set_error("pt_set_meshes: mesh %d is empty or malformed", i);
            return PT_ERR_ARGUMENT;
        }
        for (int k = 0; k < 3 * m.ntriangles; ++k)
            if (m.indices[k] < 0 || m.indices[k] >= m.nvertices) { pth::set_error("pt_set_meshes: mesh %d: vertex index %d out of range (%d vertices)", i, m.indices[k], m.nvertices); return PT_ERR_ARGUMENT; }
        pth::HostMesh hm;
        hm.geom_index = m.geom_index;
        hm.v.assign(m.vertices, m.vertices + (size_t)3 * m.nvertices);
        hm.idx.assign(m.indices, m.indices + (size_t)3 * m.ntriangles);
        copy.push_back(std::move(hm));
    }
    c->meshes.swap(copy);
    for (pt_context *s : c->subs) { int rc = pt_set_meshes(s, meshes, nmeshes); if (rc) return rc; }
    return PT_OK;
}

int pt_upload_scene(pt_context *c, const pt_geom *geoms, int G, const pt_material *mats, int M, const pt_camera *cam) {
    if (!c || !geoms || !mats || !cam || G < 1 || M < 1) { pth::set_error("pt_upload_scene: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) {
        HIPCHK(hipSetDevice(c->cfg.device));
        for (pt_context *s : c->subs) { int rc = pt_upload_scene(s, geoms, G, mats, M, cam); if (rc) return rc; }
        const int W = c->subs[0]->W, H = c->subs[0]->H;
        if (c->image_own && (W != c->W || H != c->H)) { (void)hipFree(c->image_own); if (c->image == c->image_own) c->image = nullptr; c->image_own = nullptr; }
        c->W = W; c->H = H; c->G = G; c->M = M;
        c->n_own = 0;
        for (pt_context *s : c->subs) c->n_own += s->n_own;
        if (!c->image_own) HIPCHK(hipMalloc(&c->image_own, (size_t)W * H * 3 * sizeof(float)));
        HIPCHK(hipMemsetAsync(c->image_own, 0, (size_t)W * H * 3 * sizeof(float), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!c->image) c->image = c->image_own;
        c->scene_ready = true;
        return multi::rebind(c);
    }
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int W = (int)cam->resolution[0], H = (int)cam->resolution[1];
    // pool indices are 32-bit with headroom for the segment padding: <= 2^28 pixels.  Above 2^24 pixels the pool's
    // pixel word has no room for the iteration slot / count-emission flag: one iteration per launch, no direct_light.
    if (W < 2 || H < 2 || (int64_t)W * H > (1ll << 28)) { pth::set_error("pt_upload_scene: resolution %dx%d unsupported (2x2 .. 2^28 pixels)", W, H); return PT_ERR_ARGUMENT; }
    const bool big_frame = (int64_t)W * H > (1ll << 24);
    if (big_frame && c->cfg.direct_light != 0 && c->cfg.mode == 0) { pth::set_error("pt_upload_scene: direct_light supports frames up to 2^24 pixels (%dx%d asked)", W, H); return PT_ERR_ARGUMENT; }
    std::vector<GeomRec> g(G);
    std::vector<MatRec> m(M);
    for (int i = 0; i < M; ++i) {
        memset(&m[i], 0, sizeof(MatRec));
        memcpy(m[i].color, mats[i].color, 12);
        m[i].emittance = mats[i].emittance;
        memcpy(m[i].spec, mats[i].specularColor, 12);
        m[i].refl = mats[i].hasReflective;
        m[i].refr = mats[i].hasRefractive;
        m[i].ior = mats[i].indexOfRefraction;
    }
    for (int i = 0; i < G; ++i) {
        if (geoms[i].materialid < 0 || geoms[i].materialid >= M) { pth::set_error("pt_upload_scene: geom %d has materialid %d (have %d materials)", i, geoms[i].materialid, M); return PT_ERR_ARGUMENT; }
        memcpy(g[i].inv, geoms[i].inverseTransform, 48);
        memcpy(g[i].xf, geoms[i].transform, 48);
        g[i].type = geoms[i].type;
        g[i].mat = geoms[i].materialid;
        g[i].inside_hits = mats[geoms[i].materialid].hasRefractive > 0.0f ? 1 : 0;
        if (geoms[i].type == 2) g[i].inside_hits = 0;     // MESH: byte offset of its triangles, set below when data is registered
        pth::world_bounds(geoms[i], &g[i]);
    }
    // MESH primitives with registered data (the others are skipped like the reference's empty branch)
    std::vector<const pth::HostMesh *> mesh_of(G, nullptr);
    bool have_mesh = false, big_mesh = false;              // big: beyond what a (ray, triangle) pair of k_path_q<MESH> indexes
    for (const pth::HostMesh &hm : c->meshes) {
        if (hm.geom_index >= G || geoms[hm.geom_index].type != 2) { pth::set_error("pt_upload_scene: mesh registered for geom %d, which is not a MESH of this scene", hm.geom_index); return PT_ERR_ARGUMENT; }
        mesh_of[hm.geom_index] = &hm;
        have_mesh = true;
        if (hm.idx.size() / 3 >= (size_t)kMeshPairTris) big_mesh = true;
        if (c->cfg.direct_light != 0 && mats[geoms[hm.geom_index].materialid].emittance > 0.0f) { pth::set_error("pt_upload_scene: direct_light does not sample emitting meshes (geom %d)", hm.geom_index); return PT_ERR_ARGUMENT; }
    }
    const int stride = c->cfg.row_stride, offset = c->cfg.row_offset;
    const int rows = offset < H ? (H - offset + stride - 1) / stride : 0;
    const uint32_t n_own = (uint32_t)rows * (uint32_t)W;
    // Always rebuild the device state: uploads are rare (once per frame), sizes depend on the scene.
    free_scene_buffers(c);
    c->W = W; c->H = H; c->G = G; c->M = M;
    c->n_own = n_own;
    c->pix_mask = big_frame ? 0xFFFFFFFFu : 0xFFFFFFu;
    c->empty = (n_own == 0u);
    if (c->empty) {
        // a shard without rows (row_offset >= H: more GPUs x streams than rows) is valid and renders nothing
        HIPCHK(hipMalloc(&c->image_own, (size_t)W * H * 3 * sizeof(float)));
        HIPCHK(hipMemsetAsync(c->image_own, 0, (size_t)W * H * 3 * sizeof(float), c->stream));
        if (!c->image) c->image = c->image_own;
        pth::camera_basis(cam, &c->cfg, &c->cam);
        c->scene_ready = true;
        return PT_OK;
    }
    c->cull = (c->cfg.culling == 0);
    c->queue = c->cull && (c->cfg.ordering == 1 || c->cfg.ordering == 2) && G <= 32 && c->cfg.mode == 0;
    c->queue_mesh = false;
    c->geom_lds = true;
    if (have_mesh) {
        // meshes: stable kernels and the typed work queues (where they share the spheres' stack)
        c->queue_mesh = c->queue;
        for (int i = 0; i < G; ++i) {
            if (!mesh_of[i]) continue;
            uint32_t tri_offset = 0;
            const std::vector<unsigned char> blob = pth::build_mesh_blob(*mesh_of[i], &tri_offset);
            void *d_blob = nullptr;
            HIPCHK(hipMalloc(&d_blob, blob.size()));
            c->d_mesh_blobs.push_back(d_blob);
            HIPCHK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
            const unsigned long long addr = (unsigned long long)(uintptr_t)d_blob;
            const uint32_t lo32 = (uint32_t)addr, hi32 = (uint32_t)(addr >> 32);
            pth::mesh_world_bounds(geoms[i], *mesh_of[i], &g[i]);
            memcpy(&g[i].bmin[3], &lo32, 4);
            memcpy(&g[i].bmax[3], &hi32, 4);
            g[i].inside_hits = (int)tri_offset;
        }
    }
    c->nee = c->cfg.direct_light != 0 && c->cfg.mode == 0;
    if (c->nee) {
        // one kernel family implements it: segmented compaction, culling, LDS tables, generation order
        if (!c->cull) {
            pth::set_error("pt_upload_scene: direct_light needs culling=0");
            return PT_ERR_ARGUMENT;
        }
        c->queue = false;
        std::vector<uint32_t> lights;
        for (int i = 0; i < G; ++i)
            if (mats[geoms[i].materialid].emittance > 0.0f) lights.push_back((uint32_t)i);
        c->nlights = (uint32_t)lights.size();
        if (lights.empty()) lights.push_back(0u);
        HIPCHK(hipMalloc(&c->d_lights, lights.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }

    // LDS budget: tables (+ the queue kernel's face frames, cull records and four queue buffers)
    uint32_t tb = tables_bytes(G, M, c->geom_lds);
    const uint32_t stage_bytes = c->queue ? q_lds_offset(G, M) - tables_bytes(G, M, true) + kWaves * kQCap * kQFields * (uint32_t)sizeof(float) : 0u;
    if (tb + stage_bytes > 160u * 1024u) {                  // table too large for LDS (about 1 100 primitives): scalar-load path
        if (c->nee) { pth::set_error("pt_upload_scene: direct_light needs the geometry table in LDS (%d primitives do not fit)", G); return PT_ERR_ARGUMENT; }
        c->geom_lds = false;
        tb = tables_bytes(G, M, false);
    }
    c->lds_bytes = tb + stage_bytes;
    // 33..256 analytic primitives with the table in LDS: two-level cluster culling (the stable kernel's WIDE variant;
    // with ordering = 2 the whole-path kernel k_path_w renders and this variant serves the parity hooks)
    c->wide = c->cull && c->geom_lds && !c->nee && !c->queue && !have_mesh && c->cfg.mode == 0 && G > 32 && G <= 256;
    // ordering = 2 with more than 32 analytic primitives: whole paths on k_path_w, ANY count (the reference's loop takes any
    // numberOfGeoms, src/raytraceKernel.cu:137-153,192-194) -- up to 256 with byte ids and the geometry table in LDS, beyond that with
    // wide ids and the geometry gathered from global memory.  Planned HERE (grid, block shape, LDS): the launch-group size and the
    // pools below depend on whether it renders; a scene whose tables leave no room for any shape stays on the per-bounce kernels.
    c->pathw = c->cull && !c->nee && !c->queue && !have_mesh && c->cfg.mode == 0 && c->cfg.ordering == 2 && G > 32 && G < (1 << 21);
    pth::GridBuild gb;
    WideLayout wl;
    memset(&wl, 0, sizeof wl);
    if (c->pathw) {
        c->wide_big = G > 256;
        if (!c->wide_big) {
            // (the smallest narrow shape needs 91 136 bytes beside the tables; at least 1 KB of grid)
            const int64_t grid_room = (int64_t)160 * 1024 - (int64_t)tables_bytes(G, M, true) - 91136;
            if (grid_room < 1024) c->pathw = false;
            else {
                pth::build_grid(g, G, c->cfg.grid_density, (size_t)(grid_room < 24 * 1024 ? grid_room : 24 * 1024), false, &gb);
                c->wide_shape = 0;
                if (wide_lds_bytes(0, G, M, gb.ga.blob_bytes) > 160u * 1024u) c->wide_shape = 1;
                if (wide_lds_bytes(c->wide_shape, G, M, gb.ga.blob_bytes) > 160u * 1024u) c->pathw = false;
            }
        } else {
            pth::build_grid(g, G, c->cfg.grid_density, 0, true, &gb);
            // the grid in LDS beside the larger shape, else beside the smaller one, else in global memory with the larger shape
            gb.ga.in_lds = 1u;
            c->wide_shape = 2;
            if (wide_lds_bytes(2, G, M, gb.ga.blob_bytes) > 160u * 1024u) c->wide_shape = 3;
            if (wide_lds_bytes(c->wide_shape, G, M, gb.ga.blob_bytes) > 160u * 1024u) { gb.ga.in_lds = 0u; c->wide_shape = 2; }
            if (wide_lds_bytes(c->wide_shape, G, M, gb.ga.in_lds ? gb.ga.blob_bytes : 0u) > 160u * 1024u) c->pathw = false;     // (thousands of materials)
        }
    }
    // two-level culling of the many-primitive variant: clusters of <= kClusterMax primitives of one type (pt_build.cpp)
    std::vector<unsigned char> cluster_blob;
    c->nbc = c->nsc = 0; c->cluster_bytes = 0;
    if (c->wide) {
        pth::ClusterBuild cb;
        if (!pth::build_clusters(g, G, 0, &cb)) c->wide = false;       // the per-lane cluster mask has 64 bits (the parity hooks then run the one-level culling)
        else {
            cluster_blob.swap(cb.blob);
            c->nbc = cb.nbc; c->nsc = cb.nsc;
            c->cluster_bytes = (uint32_t)cluster_blob.size();
            c->lds_bytes += c->cluster_bytes;
        }
    }
    {
        const SegVariant v = {c->geom_lds, c->cull, c->nee, c->wide};
        int occ = 0;
        if (c->queue) HIPCHK(queue_setup(c->queue_mesh, c->lds_bytes, &occ));
        else HIPCHK(seg_setup(v, c->lds_bytes, &occ));
        c->occ_bounce = occ;
    }

    // persistent grid: CUs x resident blocks per CU
    int per_cu = c->cfg.blocks_per_cu;
    if (per_cu <= 0) per_cu = c->occ_bounce;
    int grid = c->n_cu * per_cu;

    {
        // segment size: by default one segment per resident wave (every wave gets equal work in the
        // first, largest bounce and no second round is needed); cfg.chunk_rays overrides.
        // Level 0: segments of S0 slots (default 64 = one full wave group).  While halving the segment
        // count still leaves about one segment per resident wave, a bounce MERGES neighbours (the
        // output level has 2S slots per segment): early bounces see many equal segments per wave
        // (balanced), late bounces see few, re-densified ones (full wave groups).
        uint32_t S = c->cfg.chunk_rays > 0 ? (uint32_t)c->cfg.chunk_rays : 192u;
        if (S < 16u) S = 16u;
        if (S > 4096u) S = 4096u;
        c->seg_slots = S;
        // iterations per launch group: explicit, or enough to put ~32 M rays into a launch (bigger
        // launches amortise the tail of the static schedule; essential when the frame is sharded over
        // GPUs).  The pool's pixel word keeps the slot in its top 8 bits.
        // (will the scene run on the whole-path kernels?  they keep no ray pools: a launch group may be bigger, and the pools below
        // only serve the one-iteration parity hooks)
        const bool whole_path = c->cfg.ordering == 2 && c->cfg.mode == 0 &&
                                (c->pathw || (c->queue && !c->nee && !big_mesh) || (c->nee && c->cull && G <= 32 && !have_mesh));
        uint32_t K = 1;
        if (c->cfg.mode == 0 && !big_frame) {
            if (c->cfg.batch > 0) K = (uint32_t)c->cfg.batch;
            else if (whole_path) K = (uint32_t)((96u * 1024u * 1024u) / n_own);     // configs[4] (4K): 20 iterations per launch 0.925, 10: 0.929, 4: 0.952 ms/step
            else K = (uint32_t)((32u * 1024u * 1024u) / n_own);      // 16 at 1080p: measured best (14: +4 %, 18: +8 % time)
            // the slot field has 7 bits beside the count-emission flag; a plane holds the owned pixels: <= 4 GiB in all
            // ... and never more than a quarter of the device memory that is free right now (several contexts and streams share a GPU)
            const uint64_t plane_bytes = (uint64_t)n_own * 3 * sizeof(float);
            uint64_t plane_budget = 4ull << 30;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b / 4u < plane_budget) plane_budget = (uint64_t)free_b / 4u;
            const uint32_t by_memory = (uint32_t)(plane_budget / plane_bytes);
            if (K > 128u) K = 128u;
            if (K > by_memory) K = by_memory;
            if (K < 1u) K = 1u;
        }
        c->batch_max = K;
        const uint32_t max_rays = (whole_path ? 1u : K) * n_own;
        c->nseg = (max_rays + S - 1) / S;                 // S here = the smallest segment size in use
        c->cap = max_rays + 2u * 4096u;
        if (K > 1u || c->nee) {
            HIPCHK(hipMalloc(&c->d_planes, (size_t)K * n_own * 3 * sizeof(float)));
            HIPCHK(hipMemsetAsync(c->d_planes, 0, (size_t)K * n_own * 3 * sizeof(float), c->stream));
        }
        const uint32_t blocks_needed = (c->nseg + kWaves - 1) / kWaves;
        if ((uint32_t)grid > blocks_needed) grid = (int)blocks_needed;
        if (grid < 1) grid = 1;
        c->grid_bounce = grid;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc(&c->d_segcnt[i], (size_t)(c->nseg + 2u) * sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(c->d_segcnt[i], 0, (size_t)(c->nseg + 2u) * sizeof(uint32_t), c->stream));
        }
    }
    if (grid < 1) grid = 1;
    c->grid_bounce = grid;

    for (int i = 0; i < 2; ++i) HIPCHK(hipMalloc(&c->pool[i], (size_t)c->cap * kFields * sizeof(float)));
    HIPCHK(hipMalloc(&c->image_own, (size_t)W * H * 3 * sizeof(float)));
    HIPCHK(hipMemsetAsync(c->image_own, 0, (size_t)W * H * 3 * sizeof(float), c->stream));
    if (!c->image) c->image = c->image_own;
    HIPCHK(hipMalloc(&c->d_geoms, (size_t)G * sizeof(GeomRec) + cluster_blob.size()));
    if (!cluster_blob.empty())
        HIPCHK(hipMemcpy(reinterpret_cast<char *>(c->d_geoms) + (size_t)G * sizeof(GeomRec), cluster_blob.data(), cluster_blob.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->d_mats, (size_t)M * sizeof(MatRec)));
    HIPCHK(hipMalloc(&c->d_display, (size_t)W * H * sizeof(uchar4)));
    HIPCHK(hipMemcpy(c->d_geoms, g.data(), (size_t)G * sizeof(GeomRec), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_mats, m.data(), (size_t)M * sizeof(MatRec), hipMemcpyHostToDevice));
    // direct_light with ordering = 2 on a scene the typed queues take (<= 32 analytic primitives, no meshes): whole paths in one launch
    c->pathq_nee = c->nee && c->cull && c->cfg.ordering == 2 && G <= 32 && !have_mesh;
    if (c->queue || c->pathq_nee) {
        // per box primitive and axis: unit normal + the two tangent frames scatter() would derive per ray; evaluated
        // here with the kernels' own functions (pt_device.hpp is host-callable, same -ffp-contract=off build)
        std::vector<FaceFrame> fr((size_t)G * 3);
        memset(fr.data(), 0, fr.size() * sizeof(FaceFrame));
        for (int i = 0; i < G; ++i)
            if (g[i].type == 1)
                for (int col = 0; col < 3; ++col) make_face_frame(g[i].xf, col, &fr[(size_t)i * 3 + col]);
        HIPCHK(hipMalloc(&c->d_frames, fr.size() * sizeof(FaceFrame)));
        HIPCHK(hipMemcpy(c->d_frames, fr.data(), fr.size() * sizeof(FaceFrame), hipMemcpyHostToDevice));
        // the culling pass's bounds, cubes first (MESH primitives have no entry: the empty branch of the reference)
        std::vector<CullRec> cr;
        auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        for (int pass = 1; pass >= 0; --pass)
            for (int i = 0; i < G; ++i) {
                const bool boxlike = g[i].type == 1 || (g[i].type == 2 && g[i].inside_hits != 0);       // cubes and meshes: an AABB
                if (pass == 1 ? !boxlike : g[i].type != 0) continue;
                CullRec r;
                memset(&r, 0, sizeof r);
                if (pass == 1) {
                    for (int k = 0; k < 3; ++k) { r.a[k] = g[i].bmin[k]; r.b[k] = g[i].bmax[k]; }
                    r.a[3] = bits((uint32_t)i); r.b[3] = bits(1u << i);
                } else {
                    for (int k = 0; k < 4; ++k) r.a[k] = g[i].bmin[k];
                    r.b[0] = bits((uint32_t)i); r.b[1] = bits(1u << i); r.b[3] = g[i].bmax[3];
                }
                cr.push_back(r);
            }
        c->q_nbox = 0; c->q_nsph = 0;
        for (int i = 0; i < G; ++i) { if (g[i].type == 1 || (g[i].type == 2 && g[i].inside_hits != 0)) c->q_nbox++; else if (g[i].type == 0) c->q_nsph++; }
        if (cr.empty()) cr.emplace_back();
        HIPCHK(hipMalloc(&c->d_cull, cr.size() * sizeof(CullRec)));
        HIPCHK(hipMemcpy(c->d_cull, cr.data(), cr.size() * sizeof(CullRec), hipMemcpyHostToDevice));
    }
    // ordering = 2: whole paths in one launch -- a persistent grid of its own and the waves' level rings
    c->pathq = (c->queue && c->cfg.ordering == 2 && !c->nee && !big_mesh) || c->pathq_nee;       // (a mesh of 2^24 triangles or more: per-bounce kernels)
    if (c->pathq) {
        // records per wave: the largest instantiated capacity that leaves the occupancy target standing (five blocks per CU) beside
        // this scene's tables (and the scratch of the mesh stages); if none does, the smallest
        int occ = 0;
        const int target = 5;
        for (int cap : kPCaps) {
            c->path_cap = cap;
            c->lds_path = c->queue_mesh ? p_mesh_lds_bytes(G, M, (uint32_t)cap) : p_lds_bytes(G, M, (uint32_t)cap);
            if (c->lds_path > 160u * 1024u) continue;
            HIPCHK(path_setup(c->queue_mesh, c->pathq_nee, cap, c->lds_path, &occ));
            // LDS is handed out in granules of 1 280 bytes (160 KB / 128): the occupancy query does not round, the hardware does --
            // 32 640 bytes per block were reported as five blocks per CU and ran as four (0.754 vs 0.591 ms/step on the mesh scene)
            const uint32_t granules = (c->lds_path + 1279u) / 1280u;
            if (occ >= target && (uint32_t)target * granules * 1280u > 160u * 1024u) occ = target - 1;
            if (occ >= target) break;
        }
        if (c->cfg.blocks_per_cu > 0) occ = c->cfg.blocks_per_cu;
        c->grid_path = c->n_cu * occ;
        c->path_waves = (uint32_t)kWaves;
        const size_t extra = c->pathq_nee ? kNeeExtraFields : 0;
        c->arena_bytes = (size_t)c->grid_path * kWaves * (((size_t)kSFields + extra) * (kStack + (c->queue_mesh ? kMStack : 0u)) + ((size_t)kPParked + extra) * (size_t)c->path_cap +
                                                           (c->queue_mesh ? (size_t)kMFields * kMStack : 0)) * sizeof(float);
        HIPCHK(hipMalloc(&c->d_arena, c->arena_bytes));
        HIPCHK(hipMalloc(&c->d_tickets, (size_t)kTicketCtrs * kTicketStride * sizeof(uint32_t)));
    }
    if (c->pathw) {
        // k_path_w -- one big block per CU; per-wave ray slots and work stacks in LDS, the survivors' stacks and the slots' payload
        // in one arena per wave (grid and block shape: planned above)
        c->grid = gb.ga;
        HIPCHK(hipMalloc(&c->d_grid, gb.blob.size()));
        HIPCHK(hipMemcpy(c->d_grid, gb.blob.data(), gb.blob.size(), hipMemcpyHostToDevice));
        c->grid.blob = c->d_grid;
        HIPCHK(wide_setup(c->wide_shape, G, M, (!c->wide_big || c->grid.in_lds) ? c->grid.blob_bytes : 0u, &wl));
        c->lds_path = wl.lds_bytes;
        c->path_waves = wl.waves_per_block;
        c->grid_path = c->n_cu;
        c->wide_stack = wl.stack_slots; c->wide_slots = wl.slots_per_wave;
        {   // candidate keys: conservative entry distance in 250 steps of the scene's diagonal; the re-check slack of the scene
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, smax = 0.0;
            for (int i = 0; i < G; ++i) {
                for (int k = 0; k < 3; ++k) {
                    const double a0 = g[i].type == 0 ? (double)g[i].bmin[k] - g[i].bmax[3] : g[i].bmin[k];
                    const double a1 = g[i].type == 0 ? (double)g[i].bmin[k] + g[i].bmax[3] : g[i].bmax[k];
                    lo[k] = std::fmin(lo[k], a0); hi[k] = std::fmax(hi[k], a1);
                }
                smax = std::fmax(smax, (double)g[i].slack);
            }
            const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
            c->wide_qscale = (float)((c->wide_big ? 1000.0 : 250.0) / (diag > 0.0 ? diag : 1.0));
            c->wide_slack = (float)smax;
        }
        c->arena_bytes = (size_t)c->grid_path * wl.waves_per_block * ((size_t)kWalkBins * kSFields * wl.stack_slots + (size_t)kWPayload * wl.payload_per_wave) * sizeof(float);
        if (c->arena_bytes >= (1ull << 32)) { pth::set_error("pt_upload_scene: k_path_w arena of %zu bytes exceeds buffer addressing", c->arena_bytes); return PT_ERR_ARGUMENT; }
        HIPCHK(hipMalloc(&c->d_arena, c->arena_bytes));
        HIPCHK(hipMalloc(&c->d_tickets, (size_t)kTicketCtrs * kTicketStride * sizeof(uint32_t)));
        // shading frames of the box primitives (global memory here: gathered per hit)
        std::vector<FaceFrame> fr((size_t)G * 3);
        memset(fr.data(), 0, fr.size() * sizeof(FaceFrame));
        for (int i = 0; i < G; ++i)
            if (g[i].type == 1)
                for (int col = 0; col < 3; ++col) make_face_frame(g[i].xf, col, &fr[(size_t)i * 3 + col]);
        HIPCHK(hipMalloc(&c->d_frames, fr.size() * sizeof(FaceFrame)));
        HIPCHK(hipMemcpy(c->d_frames, fr.data(), fr.size() * sizeof(FaceFrame), hipMemcpyHostToDevice));
    }
    pth::camera_basis(cam, &c->cfg, &c->cam);
    c->scene_ready = true;
    return PT_OK;
}

int pt_set_image(pt_context *c, const float *host_rgb) {
    if (!c || !c->scene_ready) { pth::set_error("pt_set_image: no scene uploaded"); return PT_ERR_STATE; }
    HIPCHK(hipSetDevice(c->cfg.device));
    for (pt_context *s : c->subs) HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t bytes = (size_t)c->W * c->H * 3 * sizeof(float);
    if (host_rgb) HIPCHK(hipMemcpy(c->image, host_rgb, bytes, hipMemcpyHostToDevice));
    else HIPCHK(hipMemsetAsync(c->image, 0, bytes, c->stream));
    if (!c->subs.empty()) HIPCHK(hipStreamSynchronize(c->stream));        // the sub-contexts' streams do not order against it
    return PT_OK;
}

int pt_bind_device_image(pt_context *c, void *device_rgb) {
    if (!c) { pth::set_error("pt_bind_device_image: null context"); return PT_ERR_ARGUMENT; }
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->image = device_rgb ? static_cast<float *>(device_rgb) : c->image_own;
    if (!c->subs.empty()) {
        for (pt_context *s : c->subs) HIPCHK(hipStreamSynchronize(s->stream));
        if (c->image) return multi::rebind(c);
    }
    return PT_OK;
}

int pt_get_image(pt_context *c, float *host_rgb) {
    if (!c || !c->scene_ready || !host_rgb) { pth::set_error("pt_get_image: bad state/argument"); return PT_ERR_STATE; }
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->subs.empty()) {
        int rc = multi::for_all(c, pt_sync);
        if (rc) return rc;
        HIPCHK(hipMemcpy(host_rgb, c->image, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return PT_OK;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(host_rgb, c->image, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return check_device_error(c);
}

int pt_get_rows(pt_context *c, float *host_rgb) {
    if (!c || !c->scene_ready || !host_rgb) { pth::set_error("pt_get_rows: bad state/argument"); return PT_ERR_STATE; }
    HIPCHK(hipSetDevice(c->cfg.device));
    int rc = pt_sync(c);
    if (rc) return rc;
    // the rows y = row_offset + k*row_stride of the (shared) full-frame image: one strided 2-D copy
    const int off = c->cfg.row_offset, stride = c->cfg.row_stride;
    if (off >= c->H) return PT_OK;
    const size_t row_bytes = (size_t)c->W * 3 * sizeof(float);
    const size_t rows = (size_t)(c->H - off + stride - 1) / stride;
    HIPCHK(hipMemcpy2D(host_rgb + (size_t)off * c->W * 3, row_bytes * stride, c->image + (size_t)off * c->W * 3, row_bytes * stride,
                       row_bytes, rows, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_gather_rows_peer(pt_context *dst, pt_context *src) {
    if (!dst || !src || !dst->scene_ready || !src->scene_ready) { pth::set_error("pt_gather_rows_peer: both contexts need a scene"); return PT_ERR_STATE; }
    if (dst->W != src->W || dst->H != src->H) { pth::set_error("pt_gather_rows_peer: resolutions differ"); return PT_ERR_ARGUMENT; }
    if (dst == src || dst->image == src->image) return PT_OK;
    int rc = pt_sync(src);
    if (rc) return rc;
    rc = pt_sync(dst);
    if (rc) return rc;
    const int off = src->cfg.row_offset, stride = src->cfg.row_stride;
    if (off >= src->H) return PT_OK;
    const size_t row_bytes = (size_t)src->W * 3 * sizeof(float);
    const size_t rows = (size_t)(src->H - off + stride - 1) / stride;
    HIPCHK(hipSetDevice(dst->cfg.device));
    if (dst->cfg.device != src->cfg.device) {
        int can = 0;
        HIPCHK(hipDeviceCanAccessPeer(&can, dst->cfg.device, src->cfg.device));
        if (can) {
            hipError_t e = hipDeviceEnablePeerAccess(src->cfg.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { pth::set_error("hipDeviceEnablePeerAccess failed: %s", hipGetErrorString(e)); return PT_ERR_HIP; }
            (void)hipGetLastError();
        }
    }
    // unified addressing: hipMemcpyDefault routes a cross-device copy over the peer link (xGMI)
    HIPCHK(hipMemcpy2DAsync(dst->image + (size_t)off * dst->W * 3, row_bytes * stride, src->image + (size_t)off * src->W * 3, row_bytes * stride,
                            row_bytes, rows, hipMemcpyDefault, dst->stream));
    HIPCHK(hipStreamSynchronize(dst->stream));
    return PT_OK;
}

int pt_render(pt_context *c, int first_iteration, int count) {
    if (!c || !c->scene_ready) { pth::set_error("pt_render: no scene uploaded"); return PT_ERR_STATE; }
    if (first_iteration < 1 || count < 0) { pth::set_error("pt_render: iterations are 1-based"); return PT_ERR_ARGUMENT; }
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->subs.empty()) {                      // enqueue on every stream before anything is awaited
        for (pt_context *s : c->subs) { int rc = pt_render(s, first_iteration, count); if (rc) return rc; }
        return PT_OK;
    }
    if (c->empty) { c->iterations += (uint64_t)count; return PT_OK; }
    for (int it = first_iteration; it < first_iteration + count; ++it) {
        if (c->cfg.mode == 1) {
            Scoped s(c, 1);
            FlatArgs f;
            memset(&f, 0, sizeof f);
            f.cam = c->cam; f.image = c->image; f.geoms = c->d_geoms; f.mats = c->d_mats; f.G = c->G; f.M = c->M;
            f.n_own = c->n_own; f.write_image = 1;
            flat_launch(c->stream, f, tables_bytes(c->G, c->M, true));
            HIPCHK(hipGetLastError());
            c->iterations++;
        } else {
            // launch groups of about equal size (a short last group would pay a whole launch's tail for a few iterations)
            const uint32_t rem = (uint32_t)(first_iteration + count - it);
            const uint32_t groups = (rem + c->batch_max - 1u) / c->batch_max;
            uint32_t b = (rem + groups - 1u) / groups;
            int rc = enqueue_iterations(c, (uint32_t)it, b, -1);
            if (rc) return rc;
            c->iterations += b;
            it += (int)b - 1;
        }
    }
    return PT_OK;
}

int pt_sync(pt_context *c) {
    if (!c) { pth::set_error("pt_sync: null context"); return PT_ERR_ARGUMENT; }
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->subs.empty()) return multi::for_all(c, pt_sync);
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = resolve_events(c);
    if (rc) return rc;
    return check_device_error(c);
}

int pt_display(pt_context *c, float scale, void *out, int out_is_device) {
    if (!c || !c->scene_ready) { pth::set_error("pt_display: no scene uploaded"); return PT_ERR_STATE; }
    if (!out) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->subs.empty()) {                      // the whole frame lives in the shared image: any sub-context can show it
        for (pt_context *s : c->subs) HIPCHK(hipStreamSynchronize(s->stream));
        return pt_display(c->subs[0], scale, out, out_is_device);
    }
    const uint32_t n = (uint32_t)c->W * c->H;
    if (!c->d_display && !out_is_device) HIPCHK(hipMalloc(&c->d_display, (size_t)n * sizeof(uchar4)));
    uchar4 *dst = out_is_device ? static_cast<uchar4 *>(out) : c->d_display;
    {
        Scoped s(c, 2);
        display_launch(c->stream, c->image, dst, n, scale);
        HIPCHK(hipGetLastError());
    }
    if (!out_is_device) {
        HIPCHK(hipMemcpyAsync(out, c->d_display, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return PT_OK;
}

#ifdef PT_CULL_STATS
int pt_debug_cull_stats(unsigned long long *out16) {       // analysis builds (-DPT_CULL_STATS): the sum over the kernel families
    for (int i = 0; i < 16; ++i) out16[i] = 0ull;
    cull_stats_seg(out16); cull_stats_queue(out16); cull_stats_path(out16); cull_stats_wide(out16);
    return 0;
}
extern "C" int pt_debug_phase_cycles(unsigned long long *out16) { phase_cycles_wide(out16); return 0; }   // k_path_w's phase clock
extern "C" int pt_debug_wide_stats(unsigned long long *out32) { stats_wide(out32); return 0; }             // k_path_w's stage statistics
#endif

// CPU-side probe of k_path_w's spatial index (no device needed): builds the grid of the scene exactly as pt_upload_scene
// does and walks `nrays` rays (o.xyz, d.xyz each) ON THE HOST with the kernel's own walk functions and flag logic.
int pt_debug_grid_probe(const pt_geom *geoms, int G, int density, const float *rays, int nrays, uint32_t *out_sets, uint32_t *out_info) {
    return pth::grid_probe(geoms, G, density, rays, nrays, out_sets, out_info);
}

int pt_debug_fan_probe(const pt_geom *geoms, int G, const float *rays, int nfans, uint32_t *out_sets, uint32_t *out_info) {
    return pth::fan_probe(geoms, G, rays, nfans, out_sets, out_info);
}

int pt_set_profiling(pt_context *c, int enabled) {
    if (!c) { pth::set_error("pt_set_profiling: null context"); return PT_ERR_ARGUMENT; }
    int rc = pt_sync(c);
    if (rc) return rc;
    c->cfg.profile = enabled ? 1 : 0;
    for (pt_context *s : c->subs) s->cfg.profile = c->cfg.profile;
    return PT_OK;
}

int pt_get_stats(pt_context *c, pt_stats *out) {
    if (!c || !out) { pth::set_error("pt_get_stats: null argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) {
        // counters add up; the streams' launches overlap, so the busy time reported is the longest stream's
        memset(out, 0, sizeof *out);
        for (pt_context *s : c->subs) {
            pt_stats p;
            int rc = pt_get_stats(s, &p);
            if (rc) return rc;
            for (int k = 0; k < 65; ++k) out->live[k] += p.live[k];
            out->emitted += p.emitted;
            out->generate_launches += p.generate_launches; out->bounce_launches += p.bounce_launches; out->display_launches += p.display_launches;
            if (p.generate_ms > out->generate_ms) out->generate_ms = p.generate_ms;
            if (p.bounce_ms > out->bounce_ms) out->bounce_ms = p.bounce_ms;
            if (p.display_ms > out->display_ms) out->display_ms = p.display_ms;
            if (p.iterations > out->iterations) out->iterations = p.iterations;
        }
        return PT_OK;
    }
    int rc = pt_sync(c);
    if (rc) return rc;
    SyncBlock h;
    HIPCHK(hipMemcpy(&h, c->d_sync, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->generate_ms = c->ms[0]; out->bounce_ms = c->ms[1]; out->display_ms = c->ms[2];
    out->generate_launches = c->launches[0]; out->bounce_launches = c->launches[1]; out->display_launches = c->launches[2];
    out->iterations = c->iterations;
    for (int k = 0; k <= c->cfg.max_depth && k < 65; ++k) out->live[k] = h.totals[k] + (c->counts_pending ? (uint64_t)h.counts[k] + h.counts_b[k] : 0);
    out->emitted = h.emitted;
    return PT_OK;
}

int pt_reset_stats(pt_context *c) {
    if (!c) { pth::set_error("pt_reset_stats: null context"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return multi::for_all(c, pt_reset_stats);
    int rc = pt_sync(c);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_sync, 0, sizeof(SyncBlock), c->stream));
    c->counts_pending = false;
    c->ms[0] = c->ms[1] = c->ms[2] = 0;
    c->launches[0] = c->launches[1] = c->launches[2] = 0;
    c->iterations = 0;
    return PT_OK;
}

int pt_get_resolution(pt_context *c, int *w, int *h, int *owned) {
    if (!c || !c->scene_ready) { pth::set_error("pt_get_resolution: no scene uploaded"); return PT_ERR_STATE; }
    if (w) *w = c->W;
    if (h) *h = c->H;
    if (owned) *owned = (int)c->n_own;
    return PT_OK;
}

// ---------------------------------------------------------------- parity hooks ---------

int pt_debug_primary_hits(pt_context *c, float *dir, int *hit, float *t, float *P, float *N) {
    if (!c || !c->scene_ready) { pth::set_error("pt_debug_primary_hits: no scene uploaded"); return PT_ERR_STATE; }
    if (!c->subs.empty()) { pth::set_error("pt_debug_primary_hits: parity hooks need streams = 1"); return PT_ERR_STATE; }
    if (c->empty) { pth::set_error("pt_debug_primary_hits: this context owns no rows"); return PT_ERR_STATE; }
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t n = (size_t)c->W * c->H;
    float *d_dir = nullptr, *d_t = nullptr, *d_P = nullptr, *d_N = nullptr;
    int *d_hit = nullptr;
    HIPCHK(hipMalloc(&d_dir, n * 12)); HIPCHK(hipMalloc(&d_P, n * 12)); HIPCHK(hipMalloc(&d_N, n * 12));
    HIPCHK(hipMalloc(&d_t, n * 4)); HIPCHK(hipMalloc(&d_hit, n * 4));
    HIPCHK(hipMemsetAsync(d_hit, 0xFF, n * 4, c->stream));
    FlatArgs f;
    memset(&f, 0, sizeof f);
    f.cam = c->cam; f.image = c->image; f.geoms = c->d_geoms; f.mats = c->d_mats; f.G = c->G; f.M = c->M;
    f.n_own = c->n_own; f.write_image = 0;
    f.dir = d_dir; f.t = d_t; f.P = d_P; f.N = d_N; f.hit = d_hit;
    flat_launch(c->stream, f, tables_bytes(c->G, c->M, true));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (dir) HIPCHK(hipMemcpy(dir, d_dir, n * 12, hipMemcpyDeviceToHost));
    if (P) HIPCHK(hipMemcpy(P, d_P, n * 12, hipMemcpyDeviceToHost));
    if (N) HIPCHK(hipMemcpy(N, d_N, n * 12, hipMemcpyDeviceToHost));
    if (t) HIPCHK(hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost));
    if (hit) HIPCHK(hipMemcpy(hit, d_hit, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_dir); (void)hipFree(d_P); (void)hipFree(d_N); (void)hipFree(d_t); (void)hipFree(d_hit);
    return PT_OK;
}

int pt_debug_trace_pool(pt_context *c, int iteration, int bounces, int *count, float *ox, float *oy, float *oz,
                        float *dx, float *dy, float *dz, float *tr, float *tg, float *tb, uint32_t *pixel) {
    if (!c || !c->scene_ready || c->cfg.mode != 0) { pth::set_error("pt_debug_trace_pool: needs a path-trace context with a scene"); return PT_ERR_STATE; }
    if (!c->subs.empty()) { pth::set_error("pt_debug_trace_pool: parity hooks need streams = 1"); return PT_ERR_STATE; }
    if (bounces < 0 || bounces > c->cfg.max_depth || iteration < 1) { pth::set_error("pt_debug_trace_pool: bad bounces/iteration"); return PT_ERR_ARGUMENT; }
    if (c->empty) { if (count) *count = 0; return PT_OK; }
    HIPCHK(hipSetDevice(c->cfg.device));
    // render into a scratch accumulator and restore the counters afterwards: the hook leaves image
    // and statistics untouched
    HIPCHK(hipStreamSynchronize(c->stream));
    SyncBlock snapshot;
    HIPCHK(hipMemcpy(&snapshot, c->d_sync, sizeof snapshot, hipMemcpyDeviceToHost));
    const bool pending = c->counts_pending;
    const uint32_t bank_saved = c->bank;
    float *saved = c->image, *scratch = nullptr;
    HIPCHK(hipMalloc(&scratch, (size_t)c->W * c->H * 3 * sizeof(float)));
    HIPCHK(hipMemsetAsync(scratch, 0, (size_t)c->W * c->H * 3 * sizeof(float), c->stream));
    // ordering = 2: the whole-path kernel itself is probed -- the rays that survive bounce `bounces` - 1 leave through a tap instead
    // of going on (pool 0 = the camera rays and the pool after the last bounce keep the per-bounce kernels: a whole path has neither)
    const bool tapped = (c->pathq || c->pathw) && !c->nee && bounces >= 1 && bounces < c->cfg.max_depth;
    // whatever happens below, the context gets its image, counter bank, counters and tap state back before this returns
    auto restore = [&]() {
        c->image = saved;
        c->tap_level = 0;
        c->bank = bank_saved;
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(c->d_sync, &snapshot, sizeof snapshot, hipMemcpyHostToDevice);
        c->counts_pending = pending;
        if (scratch) { (void)hipFree(scratch); scratch = nullptr; }
        if (c->d_tap) { (void)hipFree(c->d_tap); c->d_tap = nullptr; }
        if (c->d_tap_count) { (void)hipFree(c->d_tap_count); c->d_tap_count = nullptr; }
    };
    if (tapped) {
        c->tap_cap = c->n_own;
        hipError_t et = hipMalloc(&c->d_tap, (size_t)c->tap_cap * 10 * sizeof(float));
        if (et == hipSuccess) et = hipMalloc(&c->d_tap_count, sizeof(uint32_t));
        if (et == hipSuccess) et = hipMemsetAsync(c->d_tap_count, 0, sizeof(uint32_t), c->stream);
        if (et != hipSuccess) { restore(); HIPCHK(et); }
        c->tap_level = (uint32_t)bounces;
    }
    c->image = scratch;
    int rc = enqueue_iterations(c, (uint32_t)iteration, 1u, tapped ? -1 : bounces);
    c->image = saved;
    c->tap_level = 0;
    if (rc) { restore(); return rc; }
    {
        const hipError_t es = hipStreamSynchronize(c->stream);
        if (es != hipSuccess) { restore(); HIPCHK(es); }
    }
    if (tapped) {
        uint32_t n = 0;
        std::vector<float> rec((size_t)c->tap_cap * 10);
        const hipError_t e1 = hipMemcpy(&n, c->d_tap_count, sizeof n, hipMemcpyDeviceToHost);
        const hipError_t e2 = hipMemcpy(rec.data(), c->d_tap, rec.size() * sizeof(float), hipMemcpyDeviceToHost);
        SyncBlock after;
        const hipError_t e3 = hipMemcpy(&after, c->d_sync, sizeof after, hipMemcpyDeviceToHost);
        const uint32_t live = c->bank ? after.counts_b[bounces] : after.counts[bounces];
        if (e3 == hipSuccess && after.error) snapshot.error = after.error;   // a guard that fired in the hook's launch stays visible
        restore();
        HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
        { const int rce = check_device_error(c); if (rce) return rce; }
        if (n > c->tap_cap || n != live) { pth::set_error("pt_debug_trace_pool: the tap holds %u rays, the live counter of bounce %d says %u", n, bounces, live); return PT_ERR_HIP; }
        // the waves met the rays in their own order: generation order = pixel order
        const size_t cap = c->tap_cap;
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        auto pix = [&](uint32_t i) { uint32_t v; memcpy(&v, &rec[9 * cap + i], 4); return v & c->pix_mask; };
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return pix(x) < pix(y); });
        float *dst[9] = {ox, oy, oz, dx, dy, dz, tr, tg, tb};
        for (int f = 0; f < 9; ++f)
            if (dst[f]) for (uint32_t i = 0; i < n; ++i) dst[f][i] = rec[(size_t)f * cap + order[i]];
        if (pixel) for (uint32_t i = 0; i < n; ++i) memcpy(&pixel[i], &rec[9 * cap + order[i]], 4);
        if (count) *count = (int)n;
        return check_device_error(c);
    }
    SyncBlock after;
    const hipError_t ea = hipMemcpy(&after, c->d_sync, sizeof after, hipMemcpyDeviceToHost);
    const bool fused = bounces != 0;
    const uint32_t n = ea == hipSuccess ? ((fused && c->bank) ? after.counts_b[bounces] : after.counts[bounces]) : 0u;
    if (ea == hipSuccess && after.error) snapshot.error = after.error;       // a guard that fired in the hook's launches stays visible
    restore();
    HIPCHK(ea);
    if (count) *count = (int)n;
    const float *src = c->pool[bounces & 1];
    float *dst[9] = {ox, oy, oz, dx, dy, dz, tr, tg, tb};
    if (n) {
        // segments are dense prefixes in generation order: concatenate them
        const uint32_t nseg = c->cur_nseg, slots = c->cur_slots;
        std::vector<uint32_t> cnt(nseg);
        HIPCHK(hipMemcpy(cnt.data(), c->d_segcnt[bounces & 1], (size_t)nseg * 4, hipMemcpyDeviceToHost));
        std::vector<float> field(c->cap);
        uint64_t total = 0;
        for (uint32_t sgi = 0; sgi < nseg; ++sgi) total += cnt[sgi];
        if (total != n) { pth::set_error("segment counts (%llu) disagree with the live counter (%u)", (unsigned long long)total, n); return PT_ERR_HIP; }
        for (int f = 0; f < 10; ++f) {
            float *out = f < 9 ? dst[f] : reinterpret_cast<float *>(pixel);
            if (!out) continue;
            HIPCHK(hipMemcpy(field.data(), src + (size_t)f * c->cap, (size_t)c->cap * 4, hipMemcpyDeviceToHost));
            size_t w = 0;
            for (uint32_t sgi = 0; sgi < nseg; ++sgi) {
                memcpy(out + w, field.data() + (size_t)sgi * slots, (size_t)cnt[sgi] * 4);
                w += cnt[sgi];
            }
        }
    }
    // direct_light: camera rays carry the count-emission flag implicitly (generation is fused into the
    // first bounce launch and never writes it); show it the way later pools do
    if (c->nee && bounces == 0 && pixel)
        for (uint32_t i = 0; i < n; ++i) pixel[i] |= 0x80000000u;
    return check_device_error(c);
}

int pt_debug_path_shape(pt_context *c, unsigned int *out8) {
    if (!c || !out8) { pth::set_error("pt_debug_path_shape: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return pt_debug_path_shape(c->subs[0], out8);
    if (!c->scene_ready) { pth::set_error("pt_debug_path_shape: no scene uploaded"); return PT_ERR_STATE; }
    const unsigned int blocks = c->n_cu > 0 ? (unsigned int)(c->grid_path / c->n_cu) : 0u;
    out8[0] = c->pathw ? 2u : c->pathq ? 1u : 0u;
    out8[1] = (c->pathq || c->pathw) ? c->path_waves : 0u;
    out8[2] = (c->pathq || c->pathw) ? blocks : 0u;
    out8[3] = (c->pathq || c->pathw) ? c->lds_path : 0u;
    out8[4] = c->pathw ? c->wide_slots : c->pathq ? (unsigned int)c->path_cap : 0u;
    out8[5] = (unsigned int)(c->arena_bytes < 0xFFFFFFFFull ? c->arena_bytes : 0xFFFFFFFFull);
    out8[6] = c->queue_mesh ? 1u : 0u;
    out8[7] = c->pathq_nee ? 1u : 0u;
    return PT_OK;
}

int pt_debug_set_turn_limit(pt_context *c, unsigned int turns) {
    if (!c) { pth::set_error("pt_debug_set_turn_limit: null context"); return PT_ERR_ARGUMENT; }
    c->turn_limit = turns;
    for (pt_context *s : c->subs) s->turn_limit = turns;
    return PT_OK;
}

int pt_debug_rng_from_thread(pt_context *c, float resx, float resy, float time, int n, const int *xy, float *out3) {
    (void)resy;
    if (!c || n < 0 || !xy || !out3) { pth::set_error("pt_debug_rng_from_thread: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return pt_debug_rng_from_thread(c->subs[0], resx, resy, time, n, xy, out3);
    if (n == 0) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    int *d_xy = nullptr; float *d_out = nullptr;
    HIPCHK(hipMalloc(&d_xy, (size_t)n * 8)); HIPCHK(hipMalloc(&d_out, (size_t)n * 12));
    HIPCHK(hipMemcpy(d_xy, xy, (size_t)n * 8, hipMemcpyHostToDevice));
    kat_rng_from_thread(c->stream, resx, time, n, d_xy, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out3, d_out, (size_t)n * 12, hipMemcpyDeviceToHost));
    (void)hipFree(d_xy); (void)hipFree(d_out);
    return PT_OK;
}

int pt_debug_hemisphere(pt_context *c, int n, const float *normal3, const float *xi2, float *out3) {
    if (!c || n < 0 || !normal3 || !xi2 || !out3) { pth::set_error("pt_debug_hemisphere: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return pt_debug_hemisphere(c->subs[0], n, normal3, xi2, out3);
    if (n == 0) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    float *d_n = nullptr, *d_x = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc(&d_n, (size_t)n * 12)); HIPCHK(hipMalloc(&d_x, (size_t)n * 8)); HIPCHK(hipMalloc(&d_o, (size_t)n * 12));
    HIPCHK(hipMemcpy(d_n, normal3, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_x, xi2, (size_t)n * 8, hipMemcpyHostToDevice));
    kat_hemisphere(c->stream, n, d_n, d_x, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out3, d_o, (size_t)n * 12, hipMemcpyDeviceToHost));
    (void)hipFree(d_n); (void)hipFree(d_x); (void)hipFree(d_o);
    return PT_OK;
}

int pt_debug_light_points(pt_context *c, int geom, int n, const float *seeds, float *out3) {
    if (c && !c->subs.empty()) return pt_debug_light_points(c->subs[0], geom, n, seeds, out3);
    if (!c || !c->scene_ready || geom < 0 || geom >= c->G || n < 0 || !seeds || !out3) { pth::set_error("pt_debug_light_points: bad argument"); return PT_ERR_ARGUMENT; }
    if (n == 0) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    float *d_s = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc(&d_s, (size_t)n * 4)); HIPCHK(hipMalloc(&d_o, (size_t)n * 12));
    HIPCHK(hipMemcpy(d_s, seeds, (size_t)n * 4, hipMemcpyHostToDevice));
    kat_light_points(c->stream, (const GeomRec *)(c->d_geoms + geom), n, d_s, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out3, d_o, (size_t)n * 12, hipMemcpyDeviceToHost));
    (void)hipFree(d_s); (void)hipFree(d_o);
    return PT_OK;
}

int pt_debug_sincos(pt_context *c, int n, const float *a, float *s, float *co) {
    if (!c || n < 0 || !a || !s || !co) { pth::set_error("pt_debug_sincos: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return pt_debug_sincos(c->subs[0], n, a, s, co);
    if (n == 0) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    float *d_a = nullptr, *d_s = nullptr, *d_c = nullptr;
    HIPCHK(hipMalloc(&d_a, (size_t)n * 4)); HIPCHK(hipMalloc(&d_s, (size_t)n * 4)); HIPCHK(hipMalloc(&d_c, (size_t)n * 4));
    HIPCHK(hipMemcpy(d_a, a, (size_t)n * 4, hipMemcpyHostToDevice));
    kat_sincos(c->stream, n, d_a, d_s, d_c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(s, d_s, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(co, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_a); (void)hipFree(d_s); (void)hipFree(d_c);
    return PT_OK;
}

}  // extern "C"
