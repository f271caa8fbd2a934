// pt_kernels.hip -- gfx950 kernels + the device half of the C ABI (include/ptmi355.h).
//
// Replaces the body of cudaRaytraceCore (/root/reference/src/raytraceKernel.cu:164-227) and the
// kernels it launches (raytraceRay :123-159, sendImageToPBO :88-119) with a wavefront design:
//
//   k_bounce_seg   (default) one wave streams its pool segments 64 rays at a time: [bounce 0: camera ray] ->
//                  conservative candidate culling -> exact reference tests -> scatter -> accumulate ->
//                  ballot/mbcnt compaction straight into the output segment (stable order).  Variants:
//                  NEE (direct light), WIDE (33..256 primitives: two-level cluster culling), meshes.
//   k_bounce_q     (ordering = 1, <= 32 primitives) the same work as two wave-private stages with LDS work
//                  queues by candidate type: every exact test and every shading runs on a full wave
//   k_generate + k_bounce   (compaction = 1) separate generation, dense pool, decoupled look-back scan
//   k_fold         adds the per-iteration accumulator planes of a launch group to the image in iteration order
//   k_flat         the reference kernel as shipped (one hit, flat colour overwrite) + primary-hit parity hook
//   k_display      sendImageToPBO
//
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
#include "pt_device.hpp"
#include "pt_host.hpp"

using namespace ptd;

namespace {

constexpr int kBlock = 256;          // 4 waves
#ifndef PT_SEG_WAVES
#define PT_SEG_WAVES 6               // min waves per SIMD asked of the register allocator for the bounce kernels:
#endif                               // 80 VGPRs, <= 8 B of scratch; measured 4 % faster than 5 (83 VGPRs), 7 spills
constexpr int kWaves = kBlock / 64;
constexpr int kFields = 10;          // ox oy oz dx dy dz tr tg tb pixel
constexpr uint32_t kSpinLimit = 1u << 22;

typedef unsigned long long u64;

// look-back granule: high 32 bits = state (0 empty, 1 block aggregate, 2 inclusive prefix),
// low 32 bits = value.  One naturally aligned 8-byte agent-scope store / load each.
constexpr u64 kStateAgg = 1ull << 32;
constexpr u64 kStatePrefix = 2ull << 32;

struct SyncBlock {                   // device-resident, one per context
    uint32_t counts[72];             // live rays entering bounce k of the CURRENT iteration (bank 0)
    uint32_t counts_b[72];           // bank 1: the fused segmented path alternates banks per iteration
    uint32_t tickets[72];            // chunk tickets per bounce
    u64 totals[72];                  // counts folded over finished iterations
    u64 emitted;                     // paths ended on an emitter
    uint32_t error;                  // look-back spin limit hit
    uint32_t pad;
};

struct GenArgs {
    CamRec cam;
    float *pool;                     // field f at pool + f*cap
    uint32_t cap;
    uint32_t n_own;                  // rays this context generates per iteration
    uint32_t iteration;
    SyncBlock *sync;
    u64 *status;
    uint32_t status_words;
    int depth;
    uint32_t *seg_cnt0;              // segmented mode: rays per segment entering bounce 0
    uint32_t nseg, seg_slots;
};

struct BounceArgs {
    const float *in;
    float *out;
    uint32_t cap;
    float *image;
    int G, M;
    SyncBlock *sync;
    u64 *status;                     // this bounce's granules [max_chunks]
    uint32_t rpt;                    // 64-ray groups per wave per chunk
    int bounce;
    uint32_t iteration;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ float mesh_test(const GeomRec *g, f3 o, f3 d, f3 &P, f3 &N);

// ------------------------------------------------------------------ nearest hit --------
// geometry loop of raytraceRay (src/raytraceKernel.cu:134-153)
template <typename GeomPtr>
__device__ __forceinline__ int nearest_hit(GeomPtr geoms, int G, f3 o, f3 d, float &tbest, f3 &P, f3 &N) {
    float maxd = 100000000000000000.0f;
    int hit = -1;
    for (int i = 0; i < G; ++i) {
        f3 p, n;
        float depth;
        const int type = geoms[i].type;
        if (type == 0) depth = sphere_test(geoms[i].inv, geoms[i].xf, o, d, p, n);
        else if (type == 1) depth = box_test(geoms[i].inv, geoms[i].xf, geoms[i].inside_hits, o, d, p, n);
        else if (type == 2 && geoms[i].inside_hits != 0) depth = mesh_test(&geoms[i], o, d, p, n);   // a registered mesh
        else continue;                                    // MESH without data: the reference's empty branch
        if (depth < maxd && depth > -PT_EPSILON) { maxd = depth; hit = i; P = p; N = n; }
    }
    tbest = maxd;
    return hit;
}

// ------------------------------------------------------------------ nearest hit, culled -
// Same RESULT as nearest_hit (the reference loop), fewer instructions: (A) a wave-uniform pass
// tests the ray against every primitive's conservative world-space AABB (approximate
// reciprocals, margins on both sides) and leaves a per-lane candidate mask; (B) each lane runs
// the EXACT reference test only on its own candidates, fetching that primitive's matrices with a
// per-lane index (this is what the LDS staging is for: 64 lanes read up to 64 different
// primitives per instruction) -- boxes first, then spheres, so that the two code paths do not
// diverge inside a wave.  A candidate whose box is entered farther than the best exact hit so far
// is skipped.  Nothing is culled that the exact test could report nearer than the winner, and ties
// go to the lower index exactly like the in-order reference loop (`depth < MAX_DEPTH`, first wins).
__device__ __forceinline__ float guarded_rcp(float x) {
    const float ax = fabsf(x);
    const float g = ax < 1e-30f ? copysignf(1e-30f, x) : x;
    return __builtin_amdgcn_rcpf(g);
}

// Cull-side arithmetic is NOT part of the bit-exact contract (it only decides which exact tests
// run), so it may use FMAs and approximate reciprocals -- behind explicit margins.
struct CullRay {
    f3 o, d, inv, noi;      // origin, direction, guarded 1/d, -(o * inv)
};

__device__ __forceinline__ CullRay make_cull_ray(f3 o, f3 d) {
    CullRay r;
    r.o = o; r.d = d;
    r.inv = mk(guarded_rcp(d.x), guarded_rcp(d.y), guarded_rcp(d.z));
    r.noi = mk(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
    return r;
}

// box: slab test against the inflated world AABB; returns false when the ray certainly misses it.
// tn = conservative entry distance (may be negative).
__device__ __forceinline__ bool cull_box(const float *bmin, const float *bmax, const CullRay &r, float &tn) {
    const float ax = __builtin_fmaf(bmin[0], r.inv.x, r.noi.x), bx = __builtin_fmaf(bmax[0], r.inv.x, r.noi.x);
    const float ay = __builtin_fmaf(bmin[1], r.inv.y, r.noi.y), by = __builtin_fmaf(bmax[1], r.inv.y, r.noi.y);
    const float az = __builtin_fmaf(bmin[2], r.inv.z, r.noi.z), bz = __builtin_fmaf(bmax[2], r.inv.z, r.noi.z);
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float t0m = t0 - __builtin_fmaf(fabsf(t0), 2e-5f, 2e-5f);       // entry shrunk
    const float t1p = t1 + __builtin_fmaf(fabsf(t1), 2e-5f, 2e-5f);       // exit grown
    tn = t0m;
    return !((t0m > t1p) || (t1p < 0.0f));                                // a false comparison keeps it
}

// sphere: bounding sphere (centre, inflated R^2 in bmin[3], inflated R in bmax[3]).  The margin on
// the perpendicular distance is relative to |oc|^2 because d is only unit to ~1e-6.
__device__ __forceinline__ bool cull_sphere(const float *bmin, const float *bmax, const CullRay &r, float &tn) {
    const f3 oc = mk(bmin[0] - r.o.x, bmin[1] - r.o.y, bmin[2] - r.o.z);
    const float b = __builtin_fmaf(oc.z, r.d.z, __builtin_fmaf(oc.y, r.d.y, oc.x * r.d.x));
    const float c2 = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
    const float perp2 = __builtin_fmaf(-b, b, c2);
    const float lim = __builtin_fmaf(c2, 2e-5f, bmin[3]);
    tn = b - bmax[3] - __builtin_fmaf(fabsf(b), 2e-5f, 2e-5f);
    return !((perp2 > lim) || (b < 0.0f && c2 > lim));
}

// ------------------------------------------------------------------ MESH primitive --------------
// Exact test of a MESH primitive: the nearest triangle by object-space t, ties to the earlier
// triangle -- the oracle's brute-force loop -- found through the mesh's threaded BVH.  The slab tests are cull-side
// arithmetic (FMAs, approximate reciprocals, margins on both sides, boxes inflated by the host); a node is skipped
// only if it is entered beyond the best hit so far.  Per-lane traversal: no stack, one node index per lane.
// A mesh's GeomRec carries the address of its blob [MeshNode x nnodes | MeshTri x ntris] in bmin[3] / bmax[3] and the
// byte offset of the triangles in inside_hits.
__device__ __forceinline__ float mesh_test(const GeomRec *g, f3 o, f3 d, f3 &P, f3 &N) {
    const float *inv = g->inv, *xf = g->xf;
    const unsigned long long base = ((unsigned long long)__float_as_uint(g->bmax[3]) << 32) | (unsigned long long)__float_as_uint(g->bmin[3]);
    const MeshNode *nodes = reinterpret_cast<const MeshNode *>(base);
    const MeshTri *tris = reinterpret_cast<const MeshTri *>(base + (unsigned long long)(uint32_t)g->inside_hits);
    const f3 ro = mul_point(inv, o);
    const f3 rd = normalize(mul_vector(inv, d));
    const CullRay cr = make_cull_ray(ro, rd);
    float best = 3.0e38f;
    int win = -1, widx = 0x7FFFFFFF;
    int node = 0;
    while (node >= 0) {
        const float4 lo = *reinterpret_cast<const float4 *>(nodes[node].bmin);      // bmin.xyz, skip
        const float4 hi = *reinterpret_cast<const float4 *>(nodes[node].bmax);      // bmax.xyz, leaf
        const float bl[3] = {lo.x, lo.y, lo.z}, bh[3] = {hi.x, hi.y, hi.z};
        float tn;
        const bool in = cull_box(bl, bh, cr, tn) && !(tn > best);
        const int skip = __float_as_int(lo.w), leaf = __float_as_int(hi.w);
        if (!in) { node = skip; continue; }
        if (leaf < 0) { node = node + 1; continue; }
        const int first = leaf & 0x7FFFFFF, cnt = (int)((uint32_t)leaf >> 27);
        for (int k = 0; k < cnt; ++k) {
            const float4 *tp = reinterpret_cast<const float4 *>(tris + first + k);
            const float4 a = tp[0], b = tp[1], c = tp[2];
            const float t = triangle_test(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), ro, rd);
            const int idx = __float_as_int(a.w);
            if (t > 0.0f && (t < best || (t == best && idx < widx))) { best = t; win = first + k; widx = idx; }
        }
        node = skip;
    }
    if (win < 0) return -1.0f;
    const float4 ngv = reinterpret_cast<const float4 *>(tris + win)[3];
    return mesh_finish(inv, xf, o, ro, rd, best, mk(ngv.x, ngv.y, ngv.z), P, N);
}

#ifdef PT_CULL_STATS
__device__ unsigned long long g_cull_stats[16];  // [8..15] typed-queue kernel: fresh groups, fresh valid lanes, box groups, box lanes, sphere groups, sphere lanes, shaded lanes, re-queued lanes
__device__ __forceinline__ void qstat(int i, unsigned long long v) { if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g_cull_stats[i], v); }
// [0..7] lock-step kernels: [0] groups, [1] box iters, [2] box active lanes, [3] sph iters, [4] sph active lanes, [5] candidates
#endif

template <bool GEOM_LDS>
__device__ __forceinline__ int nearest_hit_culled(const GeomRec *lg, const GeomRec *__restrict__ gg, int G, f3 o, f3 d,
                                                  float &tbest, f3 &P, f3 &N) {
    const GeomRec *tab = GEOM_LDS ? lg : gg;
    const CullRay cr = make_cull_ray(o, d);
    float best = 100000000000000000.0f;
    int hit = -1;
    for (int base = 0; base < G; base += 32) {
        const int n = (G - base) < 32 ? (G - base) : 32;
        uint32_t mask = 0u, boxbits = 0u, sphbits = 0u, meshbits = 0u;
        // nearest candidate (smallest conservative entry distance) per type: tested first, so that its
        // exact hit lets the cheap re-check below drop the lane's other candidates
        float near_t[3] = {3.0e38f, 3.0e38f, 3.0e38f};
        int near_j[3] = {-1, -1, -1};
        for (int j = 0; j < n; ++j) {                     // wave-uniform index: broadcast / scalar loads
            const GeomRec &g = tab[base + j];
            const int type = g.type;
            float tn;
            bool keep;
            int ty;
            if (type == 1) { boxbits |= 1u << j; keep = cull_box(g.bmin, g.bmax, cr, tn); ty = 0; }
            else if (type == 0) { sphbits |= 1u << j; keep = cull_sphere(g.bmin, g.bmax, cr, tn); ty = 1; }
            else if (type == 2 && g.inside_hits != 0) { meshbits |= 1u << j; keep = cull_box(g.bmin, g.bmax, cr, tn); ty = 2; }
            else continue;                                // MESH without data: empty branch in the reference
            if (keep) {
                mask |= 1u << j;
                if (tn < near_t[ty]) { near_t[ty] = tn; near_j[ty] = j; }
            }
        }
#ifdef PT_CULL_STATS
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_cull_stats[0], 1ull);
        atomicAdd(&g_cull_stats[5], (unsigned long long)__popc(mask));
#endif
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 2 && meshbits == 0u) break;       // wave-uniform: scenes without meshes never enter the pass
            uint32_t m = mask & (pass == 0 ? boxbits : pass == 1 ? sphbits : meshbits);
            bool first = true;
            while (m) {                                   // per-lane loop; the wave runs until all lanes are done
#ifdef PT_CULL_STATS
                if (pass < 2) {
                    const unsigned long long act = __ballot(1);
                    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) {
                        atomicAdd(&g_cull_stats[1 + 2 * pass], 1ull);
                        atomicAdd(&g_cull_stats[2 + 2 * pass], (unsigned long long)__popcll(act));
                    }
                }
#endif
                const int j = first ? near_j[pass] : __builtin_ctz(m);
                first = false;
                m &= ~(1u << j);
                const GeomRec *g = tab + base + j;        // per-lane gather
                if (hit >= 0) {                           // entered farther than the best exact hit: cannot win or tie
                    float tn;
                    if (pass != 1) (void)cull_box(g->bmin, g->bmax, cr, tn);
                    else (void)cull_sphere(g->bmin, g->bmax, cr, tn);
                    if (tn - g->slack > best) continue;
                }
                f3 p, nn;
                float depth;
                if (pass == 0) depth = box_test(g->inv, g->xf, g->inside_hits, o, d, p, nn);
                else if (pass == 1) depth = sphere_test(g->inv, g->xf, o, d, p, nn);
                else depth = mesh_test(g, o, d, p, nn);
                const int idx = base + j;
                if (depth > -PT_EPSILON && (depth < best || (depth == best && idx < hit))) {
                    best = depth; hit = idx; P = p; N = nn;
                }
            }
        }
    }
    tbest = best;
    return hit;
}

// Many-primitive variant (33..256 primitives).  nearest_hit_culled runs its two exact-test loops once per
// block of 32 primitives, so a 256-primitive scene pays ~13 mostly empty lock-step rounds per 64-ray group, and a
// wave-uniform scan of 256 bounds costs 256 x 22 instructions per group whatever the rays do.  Here the culling is
// two-level: the host sorts the primitives of each type into spatial clusters of 4..8 members (median splits of
// the centres); (1) a wave-uniform pass tests the <= 64 cluster boxes and leaves a 64-bit per-lane cluster mask,
// (2) each lane walks ITS clusters -- cubes first, then spheres, so that the wave stays on one code path -- and
// tests the members' own bounds through a per-lane gather from the LDS table, appending candidates to two packed
// per-lane lists (8 bits per entry, up to 8 entries per type, nearest candidate moved to the front).  The exact
// loops then run ONCE over the lists.  A wave in which any lane has more than 8 candidates of a type takes the
// brute-force reference loop, so the result is always the reference's.  Cluster boxes are unions of the members' conservative
// bounds, so a primitive the exact test can hit is always reached.
#ifndef PT_CLUSTER
#define PT_CLUSTER 4                                  // preferred members per cluster; the host grows it until <= 64 clusters
#endif
constexpr int kClusterMax = 8;
struct __attribute__((aligned(16))) ClusterRec {      // lives behind the geometry table in LDS, cube clusters first
    float bmin[3]; int first;                         // first member in the id list
    float bmax[3]; int count;
};

template <bool GEOM_LDS>
__device__ __forceinline__ int nearest_hit_wide(const GeomRec *lg, const GeomRec *__restrict__ gg, int G, int nbc, int nsc,
                                                f3 o, f3 d, float &tbest, f3 &P, f3 &N) {
    const GeomRec *tab = GEOM_LDS ? lg : gg;
    const ClusterRec *cl = reinterpret_cast<const ClusterRec *>(lg + G);
    const unsigned char *members = reinterpret_cast<const unsigned char *>(cl + nbc + nsc);
    const CullRay cr = make_cull_ray(o, d);
    // (1) wave-uniform: cluster boxes
    u64 cm = 0ull;
    for (int c = 0; c < nbc + nsc; ++c) {
        float tn;
        if (cull_box(cl[c].bmin, cl[c].bmax, cr, tn)) cm |= 1ull << c;
    }
    // (2) per lane: members of the lane's clusters -> packed candidate lists
    u64 list[2] = {0ull, 0ull};
    uint32_t cnt[2] = {0u, 0u}, near_pos[2] = {0u, 0u};
    float near_t[2] = {3.0e38f, 3.0e38f};
    bool overflow = false;
    const u64 boxclusters = nbc >= 64 ? ~0ull : ((1ull << nbc) - 1ull);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        u64 m = pass == 0 ? (cm & boxclusters) : (cm & ~boxclusters);
        while (m) {                                       // per-lane trip count; the wave runs until all lanes are done
#ifdef PT_CULL_STATS
            { const unsigned long long act = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) { atomicAdd(&g_cull_stats[6], 1ull); atomicAdd(&g_cull_stats[7], (unsigned long long)__popcll(act)); } }
#endif
            const int c = __builtin_ctzll(m);
            m &= m - 1ull;
            const int first = cl[c].first, count = cl[c].count;      // per-lane LDS reads
#pragma unroll 1
            for (int k = 0; k < count; ++k) {
                const uint32_t p = members[first + k];
                const GeomRec *g = tab + p;               // per-lane gather of the member's own bound
                float tn;
                const bool keep = pass == 0 ? cull_box(g->bmin, g->bmax, cr, tn) : cull_sphere(g->bmin, g->bmax, cr, tn);
                if (keep) {
                    const uint32_t pos = cnt[pass];
                    if (pos < 8u) list[pass] |= (u64)p << (8u * pos); else overflow = true;
                    if (tn < near_t[pass]) { near_t[pass] = tn; near_pos[pass] = pos; }
                    cnt[pass] = pos + 1u;
                }
            }
        }
    }
    if (__any(overflow)) return nearest_hit(tab, G, o, d, tbest, P, N);      // rare: the reference loop itself (brute force)
    float best = 100000000000000000.0f;
    int hit = -1;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // nearest candidate to the front: tested first, its exact hit lets the re-check drop the others
        u64 L = list[pass];
        {
            const uint32_t sh = 8u * near_pos[pass];
            const u64 e0 = L & 0xFFull, en = (L >> sh) & 0xFFull;
            L = (L & ~(0xFFull << sh)) | (e0 << sh);
            L = (L & ~0xFFull) | en;
        }
        const uint32_t total = cnt[pass];
#ifdef PT_CULL_STATS
        if (pass == 0) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_cull_stats[0], 1ull); atomicAdd(&g_cull_stats[5], (unsigned long long)(cnt[0] + cnt[1])); }
#endif
        for (uint32_t i = 0; i < total; ++i) {            // per-lane trip count; the wave runs until all lanes are done
#ifdef PT_CULL_STATS
            { const unsigned long long act = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) { atomicAdd(&g_cull_stats[1 + 2 * pass], 1ull); atomicAdd(&g_cull_stats[2 + 2 * pass], (unsigned long long)__popcll(act)); } }
#endif
            const int p = (int)((L >> (8u * i)) & 0xFFull);
            const GeomRec *g = tab + p;                   // per-lane gather
            if (hit >= 0) {                               // entered farther than the best exact hit: cannot win or tie
                float tn;
                if (pass == 0) (void)cull_box(g->bmin, g->bmax, cr, tn);
                else (void)cull_sphere(g->bmin, g->bmax, cr, tn);
                if (tn - g->slack > best) continue;
            }
            f3 pp, nn;
            const float depth = pass == 0 ? box_test(g->inv, g->xf, g->inside_hits, o, d, pp, nn)
                                          : sphere_test(g->inv, g->xf, o, d, pp, nn);
            if (depth > -PT_EPSILON && (depth < best || (depth == best && p < hit))) {
                best = depth; hit = p; P = pp; N = nn;
            }
        }
    }
    tbest = best;
    return hit;
}

// Dynamic LDS layout (all scratch lives in the dynamic region so that its base stays 16-byte
// aligned): [0,64) control words | material table | geometry table (LDS path) | ray stage.
constexpr uint32_t kCtrlBytes = 512;         // 128 control words: [0..1] block sums, [2..17] merged drain, [18..31] parked launch constants, [32..96] k_path_q survivors per level

__device__ __forceinline__ void stage_tables(char *smem_base, const GeomRec *geoms, int G, const MatRec *mats, int M,
                                             bool geoms_in_lds, GeomRec *&lg, MatRec *&lm, uint32_t extra_bytes = 0u) {
    char *smem = smem_base + kCtrlBytes;
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem);
    lm = reinterpret_cast<MatRec *>(smem);
    const uint32_t mwords = (uint32_t)M * (sizeof(MatRec) / 4);
    const uint32_t *msrc = reinterpret_cast<const uint32_t *>(mats);
    for (uint32_t i = threadIdx.x; i < mwords; i += blockDim.x) dst[i] = msrc[i];
    lg = reinterpret_cast<GeomRec *>(smem + ((mwords * 4 + 15) & ~15u));
    if (geoms_in_lds) {
        uint32_t *gdst = reinterpret_cast<uint32_t *>(lg);
        const uint32_t gwords = (uint32_t)G * (sizeof(GeomRec) / 4) + extra_bytes / 4u;    // + cluster table behind the records
        const uint32_t *gsrc = reinterpret_cast<const uint32_t *>(geoms);
        for (uint32_t i = threadIdx.x; i < gwords; i += blockDim.x) gdst[i] = gsrc[i];
    }
    __syncthreads();
}

__host__ __device__ inline uint32_t tables_bytes(int G, int M, bool geoms_in_lds) {
    uint32_t b = kCtrlBytes + (((uint32_t)M * sizeof(MatRec) + 15) & ~15u);
    if (geoms_in_lds) b += (uint32_t)G * sizeof(GeomRec);
    return (b + 15) & ~15u;
}

// ------------------------------------------------------------------ generate -----------
__global__ __launch_bounds__(kBlock) void k_generate(GenArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    // reset the sync state of this iteration (visible to the bounce kernels at the kernel boundary)
    if (gid < a.status_words) a.status[gid] = 0ull;
    if (a.seg_cnt0 && gid < a.nseg) {
        const uint32_t first = gid * a.seg_slots;
        a.seg_cnt0[gid] = first >= a.n_own ? 0u : (a.n_own - first < a.seg_slots ? a.n_own - first : a.seg_slots);
    }
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        const uint32_t k = threadIdx.x;
        a.sync->totals[k] += a.sync->counts[k];          // fold the previous iteration (stats)
        a.sync->counts[k] = (k == 0) ? a.n_own : 0u;
        a.sync->tickets[k] = 0u;
    }
    if (gid >= a.n_own) return;
    // row-interleaved ownership: local row lr -> global row lr*stride + offset
    const uint32_t W = (uint32_t)a.cam.W;
    const uint32_t lr = gid / W, x = gid - lr * W;
    const uint32_t y = lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset;
    const uint32_t pixel = y * W + x;
    f3 o, d;
    camera_ray(a.cam, pixel, a.iteration, o, d);
    float *p = a.pool + gid;
    const size_t cap = a.cap;
    p[0 * cap] = o.x; p[1 * cap] = o.y; p[2 * cap] = o.z;
    p[3 * cap] = d.x; p[4 * cap] = d.y; p[5 * cap] = d.z;
    p[6 * cap] = 1.0f; p[7 * cap] = 1.0f; p[8 * cap] = 1.0f;
    reinterpret_cast<uint32_t *>(p)[9 * cap] = pixel;
}

// ------------------------------------------------------------------ one ray, one bounce -
// nearest hit -> material -> scatter / emit.  Returns true while the path stays alive; o, d, thr
// are updated in place.  LAST: depth exhausted -- only emitters matter, survivors are counted.
// material -> scatter / emit for a ray whose nearest hit is known.  Returns true while the path stays
// alive; o, d, thr are updated in place.  LAST: depth exhausted -- only emitters matter.
// `acc` is the frame (index = global pixel) or, with `cam` given, an accumulator plane of the owned rows only.
template <bool LAST>
__device__ __forceinline__ bool shade_hit(const MatRec m, f3 P, f3 N, int bounce, uint32_t iteration, float *image,
                                          uint32_t pixel, f3 &o, f3 &d, f3 &thr, uint32_t &emitted, const CamRec *plane_cam = nullptr) {
    if (LAST && !(m.emittance > 0.0f)) return true;       // depth exhausted: alive, contributes 0
    uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + (uint32_t)bounce));
    st = lcg_next(st); const float u_sel = u01(st);
    st = lcg_next(st); const float xi1 = u01(st);
    st = lcg_next(st); const float xi2 = u01(st);
    f3 L = mk(0.0f, 0.0f, 0.0f);
    const int code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, L);
    if (code == 3) {
        // exactly one live path per pixel per iteration (slot): plain read-modify-write, no atomics
        float *px = image + (size_t)(plane_cam ? owned_index(*plane_cam, pixel) : pixel) * 3;
        px[0] = px[0] + L.x; px[1] = px[1] + L.y; px[2] = px[2] + L.z;
        emitted++;
    }
    return code <= 2;
}

template <bool GEOM_LDS, bool LAST, bool CULL, bool WIDE = false>
__device__ __forceinline__ bool bounce_ray(const GeomRec *lg, const GeomRec *__restrict__ geoms, const MatRec *lm,
                                           int G, int bounce, uint32_t iteration, float *image, uint32_t pixel,
                                           f3 &o, f3 &d, f3 &thr, uint32_t &emitted, int nbc = 0, int nsc = 0, const CamRec *plane_cam = nullptr) {
    float t;
    f3 P, N;
    int hit;
    if (CULL && WIDE) hit = nearest_hit_wide<GEOM_LDS>(lg, geoms, G, nbc, nsc, o, d, t, P, N);
    else if (CULL) hit = nearest_hit_culled<GEOM_LDS>(lg, geoms, G, o, d, t, P, N);
    else if (GEOM_LDS) hit = nearest_hit(lg, G, o, d, t, P, N);
    else hit = nearest_hit(geoms, G, o, d, t, P, N);
    if (hit < 0) return false;
    const int mid = GEOM_LDS ? lg[hit].mat : geoms[hit].mat;
    return shade_hit<LAST>(lm[mid], P, N, bounce, iteration, image, pixel, o, d, thr, emitted, plane_cam);
}

// direct_light variant (k_bounce_seg only; LDS tables, culling on).  At a diffuse
// hit one shadow ray goes to a point on a random emitter (the reference's getRandomPointOnCube/Sphere);
// an emitter hit adds its radiance only while `flag` (camera ray / last event specular) is set.  All of
// a path's contributions land in its iteration's accumulator plane in bounce order.
template <bool LAST>
__device__ __forceinline__ bool bounce_ray_nee(const GeomRec *lg, const GeomRec *__restrict__ geoms, const MatRec *lm, int G,
                                               const uint32_t *__restrict__ lights, uint32_t nlights, int bounce,
                                               uint32_t iteration, float *acc, uint32_t acc_index, uint32_t pixel, f3 &o, f3 &d, f3 &thr,
                                               uint32_t &emitted, uint32_t &flag) {
    float t;
    f3 P, N;
    const int hit = nearest_hit_culled<true>(lg, geoms, G, o, d, t, P, N);
    if (hit < 0) return false;
    const MatRec m = lm[lg[hit].mat];
    if (LAST && !(m.emittance > 0.0f)) return true;
    uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + (uint32_t)bounce));
    st = lcg_next(st); const float u_sel = u01(st);
    st = lcg_next(st); const float xi1 = u01(st);
    st = lcg_next(st); const float xi2 = u01(st);
    const f3 d_in = d;
    f3 L = mk(0.0f, 0.0f, 0.0f);
    const int code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, L);
    float *px = acc + (size_t)acc_index * 3;
    if (code == 3) {
        if (flag) { px[0] = px[0] + L.x; px[1] = px[1] + L.y; px[2] = px[2] + L.z; }
        emitted++;
    }
    if (!LAST && code == 0 && nlights > 0u) {
        st = lcg_next(st); const float u_l = u01(st);
        st = lcg_next(st); const float seedf = (float)(st & 0xFFFFFFu);
        int li = (int)(u_l * (float)nlights);
        if (li > (int)nlights - 1) li = (int)nlights - 1;
        const int lid = (int)lights[li];
        f3 Q;
        float invpdf;
        const bool ok = sample_light(lg[lid].xf, lg[lid].type, seedf, Q, invpdf);
        const f3 wv = Q - o;
        const float dist2 = dot(wv, wv);
        if (ok && dist2 > 0.0f) {
            const float dist = __builtin_sqrtf(dist2);
            const f3 w = wv * (1.0f / dist);
            const f3 n = N * (1.0f / __builtin_sqrtf(dot(N, N)));
            const f3 nf = (dot(n, d_in) > 0.0f) ? neg(n) : n;
            const float cos_s = dot(nf, w);
            if (cos_s > 0.0f) {
                float th = 0.0f;
                f3 Ph = mk(0.0f, 0.0f, 0.0f), Nh = mk(0.0f, 0.0f, 0.0f);
                const int h = nearest_hit_culled<true>(lg, geoms, G, o, w, th, Ph, Nh);
                const float tol = 1e-3f * (dist > 1.0f ? dist : 1.0f);
                const float nl2 = dot(Nh, Nh);
                if (h == lid && (th + tol >= dist) && nl2 > 0.0f) {       // a nearer face of the same emitter hides Q
                    const float cos_l = fabsf(dot(Nh, w)) / __builtin_sqrtf(nl2);
                    const float geomf = (((cos_s * cos_l) * invpdf) / (PT_PI * dist2)) * (float)nlights;
                    const MatRec ml = lm[lg[lid].mat];
                    const f3 Le = mk(ml.color[0], ml.color[1], ml.color[2]) * ml.emittance;
                    const f3 C = (thr * Le) * geomf;
                    px[0] = px[0] + C.x; px[1] = px[1] + C.y; px[2] = px[2] + C.z;
                }
            }
        }
    }
    flag = (code == 1 || code == 2) ? 1u : 0u;
    return code <= 2;
}

// ------------------------------------------------------------------ bounce -------------
// decoupled look-back over the chunk granules; one lane.  Returns the exclusive prefix.
__device__ __forceinline__ uint32_t lookback(u64 *status, uint32_t chunk, uint32_t total, SyncBlock *sync) {
    if (chunk == 0) {
        __hip_atomic_store(&status[0], kStatePrefix | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0u;
    }
    __hip_atomic_store(&status[chunk], kStateAgg | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0u;
    uint32_t j = chunk;
    uint32_t spins = 0u;
    while (j > 0u) {
        const u64 s = __hip_atomic_load(&status[j - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t state = (uint32_t)(s >> 32);
        if (state == 0u) {
            if (++spins > kSpinLimit) { sync->error = 1u; break; }   // bounded: never hang the device
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += (uint32_t)s;
        if (state == 2u) break;
        --j;
    }
    __hip_atomic_store(&status[chunk], kStatePrefix | (u64)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

template <bool GEOM_LDS, bool LAST>
__global__ __launch_bounds__(kBlock) void k_bounce(BounceArgs a, const GeomRec *__restrict__ geoms,
                                                   const MatRec *__restrict__ mats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_chunk = ctrl;            // [2], alternating per chunk
    uint32_t *s_excl = ctrl + 2;
    uint32_t *s_wtot = ctrl + 4;         // [kWaves]

    GeomRec *lg;
    MatRec *lm;
    stage_tables(smem, geoms, a.G, mats, a.M, GEOM_LDS, lg, lm);
    const uint32_t tb = tables_bytes(a.G, a.M, GEOM_LDS);
    const uint32_t rpt = a.rpt;
    const uint32_t span = 64u * rpt;                      // rays per wave per chunk
    const uint32_t chunk_rays = span * kWaves;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    float *stage = reinterpret_cast<float *>(smem + tb) + (size_t)wave * span * kFields;   // wave-private

    const uint32_t n_in = a.sync->counts[a.bounce];
    const size_t cap = a.cap;
    uint32_t emitted = 0u, survivors_last = 0u, parity = 0u;

    for (;;) {
        if (threadIdx.x == 0) s_chunk[parity] = atomicAdd(&a.sync->tickets[a.bounce], 1u);
        __syncthreads();
        const uint32_t chunk = s_chunk[parity];
        parity ^= 1u;
        const uint32_t base = chunk * chunk_rays;
        if (base >= n_in) break;                          // uniform: every block ends here

        uint32_t wtotal = 0u;
        for (uint32_t sub = 0; sub < rpt; ++sub) {
            const uint32_t idx = base + wave * span + sub * 64u + lane;
            bool alive = false;
            f3 o, d, thr;
            uint32_t pixel = 0u;
            if (idx < n_in) {
                const float *p = a.in + idx;
                o = mk(p[0 * cap], p[1 * cap], p[2 * cap]);
                d = mk(p[3 * cap], p[4 * cap], p[5 * cap]);
                thr = mk(p[6 * cap], p[7 * cap], p[8 * cap]);
                pixel = reinterpret_cast<const uint32_t *>(p)[9 * cap];
                alive = bounce_ray<GEOM_LDS, LAST, false>(lg, geoms, lm, a.G, a.bounce, a.iteration, a.image, pixel, o, d, thr, emitted);
            }
            const u64 ballot = __ballot(alive);
            if (LAST) {
                survivors_last += (uint32_t)__popcll(ballot);
            } else {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
                if (alive) {
                    float *q = stage + wtotal + rank;
                    q[0 * span] = o.x; q[1 * span] = o.y; q[2 * span] = o.z;
                    q[3 * span] = d.x; q[4 * span] = d.y; q[5 * span] = d.z;
                    q[6 * span] = thr.x; q[7 * span] = thr.y; q[8 * span] = thr.z;
                    q[9 * span] = __uint_as_float(pixel);
                }
                wtotal += (uint32_t)__popcll(ballot);
            }
        }
        if (LAST) continue;                               // nothing is written back at the last bounce

        if (lane == 0) s_wtot[wave] = wtotal;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
            const uint32_t excl = lookback(a.status, chunk, total, a.sync);
            *s_excl = excl;
            if (base + chunk_rays >= n_in) a.sync->counts[a.bounce + 1] = excl + total;   // last chunk
        }
        __syncthreads();
        uint32_t off = *s_excl;
        for (uint32_t w = 0; w < wave; ++w) off += s_wtot[w];
        float *outp = a.out + off;
        for (uint32_t j = lane; j < wtotal; j += 64u) {
#pragma unroll
            for (int f = 0; f < kFields; ++f) outp[(size_t)f * cap + j] = stage[f * span + j];
        }
    }

    // per-block stats: one atomic each at exit
    if (LAST) {                                           // survivors_last is wave-uniform (ballot counts)
        if (lane == 0 && survivors_last) atomicAdd(&a.sync->counts[a.bounce + 1], survivors_last);
    }
    for (int s = 32; s > 0; s >>= 1) emitted += __shfl_down(emitted, s);
    if (lane == 0 && emitted) atomicAdd(&a.sync->emitted, (u64)emitted);
}

// ------------------------------------------------------------------ bounce, segmented ---
// Wave-autonomous segmented compaction (the default).  The pool is cut into fixed segments of
// S = 64*rpt slots; segment s holds cnt_in[s] live rays packed at its start, in generation
// order.  ONE WAVE owns a segment for the whole launch: it streams the segment 64 rays at a
// time, and survivors go straight from registers to the same segment of the output pool at
// base + running + mbcnt(ballot) -- no inter-wave traffic, no barrier, no ticket, no look-back.
// Global order is still generation order (segments are ordered, each is dense), so the stream
// stays coherent and the result is bit-identical to the look-back variant.
struct SegArgs {
    const float *in;
    float *out;
    uint32_t cap;
    float *image;
    int G, M;
    SyncBlock *sync;
    const uint32_t *cnt_in;          // [nseg_in]
    uint32_t *cnt_out;               // [nseg_out]
    uint32_t nseg_in, nseg_out;      // nseg_out = merge ? ceil(nseg_in/2) : nseg_in
    uint32_t seg_slots;              // S of the INPUT level (output level: merge ? 2S : S)
    uint32_t merge;                  // 1: a wave reads the pair (2j, 2j+1) and writes ONE segment at 2j*S
    int bounce;
    uint32_t iteration;
    uint32_t n_own;                  // GEN: rays of bounce 0 come from the camera, not from the pool
    uint32_t n_rays;                 // = batch * n_own: `batch` consecutive iterations share one launch
    uint32_t batch;                  // ray id = slot * n_own + local; the pool's pixel word is slot<<24 | pixel
    uint32_t pool_bytes;             // queue kernel: size of one pool in bytes when it is below 4 GiB (buffer addressing: one
                                     //   32-bit lane offset + a scalar field offset per access), else 0 (64-bit flat addresses)
    uint32_t pix_mask;               // 0xFFFFFF while the pixel word carries slot/flag bits; 0xFFFFFFFF for frames above
                                     //   2^24 pixels (then batch == 1, no direct_light: the word is the raw pixel index)
    float *planes;                   // batch > 1: one accumulator plane per in-flight iteration slot
    size_t plane_stride;             //   (floats); folded into the image in iteration order afterwards
    uint32_t bank;                   // counter bank of this launch group (the host alternates 0/1)
    const uint32_t *lights;          // direct_light: indices of the emitting primitives, in index order
    uint32_t nlights;
    int nbc, nsc;                    // many-primitive variant: cube / sphere clusters behind the geometry table
    uint32_t cluster_bytes;          //   and the size of that table (clusters + member ids, multiple of 16)
    CamRec cam;
};

template <bool GEOM_LDS, bool LAST, bool CULL, bool GEN, bool NEE = false, bool WIDE = false>
__global__ __launch_bounds__(kBlock, (NEE || WIDE) ? 4 : PT_SEG_WAVES) void k_bounce_seg(SegArgs a, const GeomRec *__restrict__ geoms,
                                                       const MatRec *__restrict__ mats) {
    static_assert(!NEE || (GEOM_LDS && CULL), "direct_light runs on the LDS tables with culling");
    static_assert(!WIDE || (GEOM_LDS && CULL && !NEE), "the many-primitive variant runs on the LDS tables with culling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);   // [0] survivors, [1] emitted (block sums)
    if (threadIdx.x < 2) ctrl[threadIdx.x] = 0u;
    GeomRec *lg;
    MatRec *lm;
    stage_tables(smem, geoms, a.G, mats, a.M, GEOM_LDS, lg, lm, WIDE ? a.cluster_bytes : 0u);     // ends with __syncthreads()

    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t wslot = blockIdx.x * kWaves + wave, nslots = gridDim.x * kWaves;
    const size_t cap = a.cap;
    const uint32_t S = a.seg_slots;
    uint32_t emitted = 0u, survivors = 0u;
    // per-iteration counter banks: this iteration adds into bank a.bank; the GEN launch
    // (first kernel of an iteration, nothing of the previous iteration still runs) folds the other
    // bank -- the previous iteration's counts -- into the running totals and clears it.
    uint32_t *bank = a.bank ? a.sync->counts_b : a.sync->counts;
    if (GEN && blockIdx.x == 0 && threadIdx.x < 72) {
        uint32_t *other = a.bank ? a.sync->counts : a.sync->counts_b;
        a.sync->totals[threadIdx.x] += other[threadIdx.x];
        other[threadIdx.x] = 0u;
        if (threadIdx.x == 0) bank[0] = a.n_rays;
    }

    for (uint32_t seg = wslot; seg < a.nseg_out; seg += nslots) {
        // input: one segment, or (merge) the two neighbours 2j and 2j+1 read as ONE list, so that the
        // wave groups stay full across the seam; output: a dense prefix starting at the first one's base
        const uint32_t sa = a.merge ? 2u * seg : seg;
        uint32_t na, nb;
        if (GEN) {                                        // level-0 segments are full except the last
            const uint32_t f0 = sa * S, f1 = f0 + S;
            na = f0 >= a.n_rays ? 0u : (a.n_rays - f0 < S ? a.n_rays - f0 : S);
            nb = (!a.merge || f1 >= a.n_rays) ? 0u : (a.n_rays - f1 < S ? a.n_rays - f1 : S);
        } else {
            na = a.cnt_in[sa];
            nb = (a.merge && sa + 1u < a.nseg_in) ? a.cnt_in[sa + 1u] : 0u;
        }
        const uint32_t n = na + nb;
        const uint32_t base = sa * S;
        uint32_t running = 0u;
        for (uint32_t g = 0; g < n; g += 64u) {
            const uint32_t k = g + lane;
            bool alive = false;
            f3 o, d, thr;
            uint32_t pixel = 0u;
            if (k < n) {
                uint32_t slot;
                uint32_t flag = 1u;                       // NEE: count-emission bit (bit 31 of the pixel word)
                if (GEN) {
                    // k_generate fused: ray id -> (iteration slot, owned pixel via the row interleave) -> camera ray
                    const uint32_t gid = base + (k < na ? k : k - na + S);
                    slot = a.batch > 1u ? gid / a.n_own : 0u;
                    const uint32_t local = gid - slot * a.n_own;
                    const uint32_t W = (uint32_t)a.cam.W;
                    const uint32_t lr = local / W, x = local - lr * W;
                    pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
                    camera_ray(a.cam, pixel, a.iteration + slot, o, d);
                    thr = mk(1.0f, 1.0f, 1.0f);
                } else {
                    // wave-uniform field bases (SGPR) + one 32-bit lane offset: `global_load_dword v, v_off, s[base]`
                    const uint32_t idx = base + (k < na ? k : k - na + S);
                    __builtin_assume(idx < (1u << 29));
                    const float *in = a.in;
                    o = mk(in[idx], (in + cap)[idx], (in + 2 * cap)[idx]);
                    d = mk((in + 3 * cap)[idx], (in + 4 * cap)[idx], (in + 5 * cap)[idx]);
                    thr = mk((in + 6 * cap)[idx], (in + 7 * cap)[idx], (in + 8 * cap)[idx]);
                    const uint32_t pv = reinterpret_cast<const uint32_t *>(in + 9 * cap)[idx];
                    slot = NEE ? (pv >> 24) & 0x7Fu : (a.batch > 1u ? pv >> 24 : 0u);
                    flag = pv >> 31;
                    pixel = pv & a.pix_mask;
                }
                if (NEE) {
                    // every contribution of a path goes to its iteration's plane (folded afterwards)
                    float *acc = a.planes + (size_t)slot * a.plane_stride;
                    alive = bounce_ray_nee<LAST>(lg, geoms, lm, a.G, a.lights, a.nlights, a.bounce, a.iteration + slot, acc, owned_index(a.cam, pixel), pixel,
                                                 o, d, thr, emitted, flag);
                    pixel |= (slot << 24) | (flag << 31);
                } else {
                    float *acc = a.batch > 1u ? a.planes + (size_t)slot * a.plane_stride : a.image;
                    alive = bounce_ray<GEOM_LDS, LAST, CULL, WIDE>(lg, geoms, lm, a.G, a.bounce, a.iteration + slot, acc, pixel, o, d, thr, emitted, a.nbc, a.nsc,
                                                                   a.batch > 1u ? &a.cam : nullptr);
                    pixel |= slot << 24;
                }
            }
            const u64 ballot = __ballot(alive);
            if (!LAST && alive) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
                const uint32_t oi = base + running + rank;
                __builtin_assume(oi < (1u << 29));
                float *out = a.out;
                out[oi] = o.x; (out + cap)[oi] = o.y; (out + 2 * cap)[oi] = o.z;
                (out + 3 * cap)[oi] = d.x; (out + 4 * cap)[oi] = d.y; (out + 5 * cap)[oi] = d.z;
                (out + 6 * cap)[oi] = thr.x; (out + 7 * cap)[oi] = thr.y; (out + 8 * cap)[oi] = thr.z;
                reinterpret_cast<uint32_t *>(out + 9 * cap)[oi] = pixel;
            }
            running += (uint32_t)__popcll(ballot);
        }
        if (!LAST && lane == 0) a.cnt_out[seg] = running;
        survivors += running;
    }

    // stats: wave sums -> block sums in LDS -> one fire-and-forget global atomic per block
    for (int sft = 32; sft > 0; sft >>= 1) emitted += __shfl_down(emitted, sft);
    if (lane == 0) {
        if (survivors) atomicAdd(&ctrl[0], survivors);
        if (emitted) atomicAdd(&ctrl[1], emitted);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ctrl[0]) atomicAdd(&bank[a.bounce + 1], ctrl[0]);
        if (ctrl[1]) atomicAdd(&a.sync->emitted, (u64)ctrl[1]);
    }
}

// ------------------------------------------------------------------ bounce, typed work queues ----
// `ordering = 1` (one launch per bounce; k_path_q below runs the same two stages over whole paths and is what bench.py
// runs).  Measured on the Cornell box (tools/qstats.py): a ray has 0.93 candidate
// primitives on average -- a third have none, most of the rest exactly one, 0.82 exact tests per ray are needed in
// all -- yet in the lock-step kernels every 64-ray group pays whole rounds of the exact cube test, the exact sphere
// test and the shading for the lanes that need them.  Here the unit of work is ONE EXACT TEST of a ray against its
// nearest candidate, and the wave regroups rays between the two stages so that both run on (nearly) full waves:
//
//   FRESH   64 rays of the wave's input stream: load origin + direction only (bounce 0: the camera ray), conservative
//           culling pass over a compact LDS table of bounds (wave-uniform index, two primitives per trip so that
//           their LDS reads overlap) -> candidate mask + nearest candidate.  Rays without candidates are finished.
//           The others are pushed -- origin, direction, pool index, remaining mask, first candidate: 9 dwords -- on
//           one of two wave-private LDS stacks by the TYPE of that nearest candidate (cubes grow up, spheres grow
//           down in one buffer).
//   TEST    pops up to 64 records of one type (a full wave whenever a stack holds 64) and runs that exact reference
//           test on all lanes.  The few rays (0.04 %) with another candidate that could still win or tie -- its
//           conservative entry distance is re-checked against the best hit -- take further rounds on the spot.  Then
//           the hits are shaded at once: throughput and pixel word are fetched from the input pool only now (four
//           dwords that finished rays never load, requested at pop time), cube hits take their unit normal and
//           tangent frame from the per-face table (FaceFrame, filled by the host with the same arithmetic), and the
//           survivors go to the wave's output stream.
//
// Output: a wave fills ITS OWN segments (seg = wave slot + k * slots) one after the other, so segments stay dense
// whatever died (no half-empty groups in late bounces).  Survivors therefore keep their wave but neither their
// segment nor their order (deterministic; image, live counts and the set of rays are those of the stable kernel --
// asserted).  Nothing leaves the wave: no barrier, no atomics on the data path; the accumulator takes memory-side
// float atomics (one addition per word and launch, i.e. the bits of a read-modify-write, without its load).
#ifndef PT_Q_CAP
#define PT_Q_CAP 152                     // records per wave, both stacks together: a FRESH group needs 64 free, so
#endif                                   // stacks of up to 88 wait; 9 dwords x 152 x 4 waves + tables -> 6 blocks / CU
constexpr uint32_t kQCap = PT_Q_CAP;
constexpr uint32_t kQFields = 9;         // ox oy oz dx dy dz idx|pixelword mask next

#ifndef PT_Q_MERGED_DRAIN
#define PT_Q_MERGED_DRAIN 1              // the four waves of a block pool their last, partly filled stacks (0: every wave drains its own)
#endif
#ifndef PT_Q_PARK
#define PT_Q_PARK 1                      // launch constants of the accumulate step live in LDS, not in scalar registers
#endif
#ifndef PT_Q_WAVES
#define PT_Q_WAVES 6
#endif
#ifdef PT_MARKERS                        // analysis builds only: named comments in the .s
#define PT_MARK(name) __asm__ volatile("; PTMARK " name)
#else
#define PT_MARK(name)
#endif

// bounds for the culling pass, sorted cubes first: 32 bytes per primitive
struct __attribute__((aligned(16))) CullRec {
    float a[4];              // cube: bmin.xyz, primitive index (int bits)   sphere: centre.xyz, inflated R^2
    float b[4];              // cube: bmax.xyz, 1 << index (int bits)        sphere: index, 1 << index, -, inflated R
};

__device__ __forceinline__ uint32_t wave_rank(u64 ballot) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

__host__ __device__ inline uint32_t q_frames_offset(int G, int M) { return tables_bytes(G, M, true); }
__host__ __device__ inline uint32_t q_cull_offset(int G, int M) { return q_frames_offset(G, M) + (uint32_t)G * 3u * (uint32_t)sizeof(FaceFrame); }
__host__ __device__ inline uint32_t q_lds_offset(int G, int M) { return q_cull_offset(G, M) + (uint32_t)G * (uint32_t)sizeof(CullRec); }

struct QTables {
    const FaceFrame *frames;     // [G][3]
    const CullRec *cull;         // [nbox + nsph], cubes first
    int nbox, nsph;
};

// MESH: the scene holds MESH primitives with triangles (their BVH traversal needs registers the common variant must not pay for)
template <bool LAST, bool GEN, bool MESH = false>
__global__ __launch_bounds__(kBlock, MESH ? 4 : PT_Q_WAVES) void k_bounce_q(SegArgs a, const GeomRec *__restrict__ geoms,
                                                                            const MatRec *__restrict__ mats, QTables qt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);   // [0] survivors, [1] emitted (block sums)
    if (threadIdx.x < 2) ctrl[threadIdx.x] = 0u;
#if PT_Q_PARK
    // Launch constants of the accumulate step, parked in LDS and read where an emitter is hit: they are needed by a few
    // lanes of a group, and as kernel arguments they would sit in scalar registers across the whole loop (the kernel
    // spills ~60 of those into vector lanes, every reload a vector instruction).
    uint32_t *park = ctrl + 18;                            // [18..31]
    if (threadIdx.x == 0) {
        const unsigned long long pl = (unsigned long long)(uintptr_t)(a.batch > 1u ? a.planes : a.image), st = (unsigned long long)a.plane_stride;
        park[0] = (uint32_t)pl; park[1] = (uint32_t)(pl >> 32); park[2] = (uint32_t)st; park[3] = (uint32_t)(st >> 32);
        park[4] = (uint32_t)a.cam.W; park[5] = (uint32_t)a.cam.row_offset; park[6] = a.cam.mW; park[7] = a.cam.shW;
        park[8] = a.cam.mS; park[9] = a.cam.shS;
    }
#endif
    GeomRec *lg;
    MatRec *lm;
    // LDS: ctrl | materials | geometry | face frames | cull records | 4 wave-private queue buffers
    FaceFrame *lf = reinterpret_cast<FaceFrame *>(smem + q_frames_offset(a.G, a.M));
    CullRec *lc = reinterpret_cast<CullRec *>(smem + q_cull_offset(a.G, a.M));
    {
        uint32_t *fd = reinterpret_cast<uint32_t *>(lf);
        const uint32_t *fs = reinterpret_cast<const uint32_t *>(qt.frames);
        for (uint32_t i = threadIdx.x; i < (uint32_t)a.G * 3u * (uint32_t)(sizeof(FaceFrame) / 4); i += blockDim.x) fd[i] = fs[i];
        uint32_t *cd = reinterpret_cast<uint32_t *>(lc);
        const uint32_t *cs = reinterpret_cast<const uint32_t *>(qt.cull);
        for (uint32_t i = threadIdx.x; i < (uint32_t)(qt.nbox + qt.nsph) * (uint32_t)(sizeof(CullRec) / 4); i += blockDim.x) cd[i] = cs[i];
    }
    stage_tables(smem, geoms, a.G, mats, a.M, true, lg, lm);        // ends with __syncthreads()

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t wslot = blockIdx.x * kWaves + wave, nslots = gridDim.x * kWaves;
    const size_t cap = a.cap;
    const uint32_t S = a.seg_slots;
    // pool accesses: with a pool below 4 GiB every load / store is `buffer_* v, v_off, s[rsrc], s_field offen` -- one 32-bit
    // lane offset per ray and a scalar offset per field instead of a 64-bit vector address per access
    const bool ub = a.pool_bytes != 0u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.pool_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.pool_bytes, 0x00020000);
    const uint32_t cap4 = (uint32_t)cap * 4u;
    // the field offset f * cap4 is multiplied out where it is used (one scalar instruction): ten products kept across
    // the loop for each pool are twenty scalar registers the kernel does not have
    auto field_off = [&](uint32_t f) -> uint32_t {
#if PT_Q_PARK
        uint32_t r;
        __asm__ volatile("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(cap4), "s"(f));
        return r;
#else
        return f * cap4;
#endif
    };
    auto ldf = [&](uint32_t f, uint32_t idx) -> float {
        return ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, idx * 4u, field_off(f), 0)) : (a.in + (size_t)f * cap)[idx];
    };
    auto stf = [&](uint32_t f, uint32_t idx, float v) {
        if (ub) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, idx * 4u, field_off(f), 0);
        else (a.out + (size_t)f * cap)[idx] = v;
    };
    uint32_t emitted = 0u, survivors = 0u;
    float *q = reinterpret_cast<float *>(smem + q_lds_offset(a.G, a.M)) + (size_t)wave * kQCap * kQFields;

    uint32_t *bank = a.bank ? a.sync->counts_b : a.sync->counts;
    if (GEN && blockIdx.x == 0 && threadIdx.x < 72) {
        uint32_t *other = a.bank ? a.sync->counts : a.sync->counts_b;
        a.sync->totals[threadIdx.x] += other[threadIdx.x];
        other[threadIdx.x] = 0u;
        if (threadIdx.x == 0) bank[0] = a.n_rays;
    }

    uint32_t boxbits = 0u, meshbits = 0u;                  // wave-uniform; G <= 32 on this path
    for (int j = 0; j < a.G; ++j) {
        if (lg[j].type == 1) boxbits |= 1u << j;
        else if (MESH && lg[j].type == 2 && lg[j].inside_hits != 0) meshbits |= 1u << j;      // a MESH with registered triangles
    }
    const uint32_t aabbbits = boxbits | meshbits;         // primitives whose conservative bound is a box

    // input cursor: the wave's segments in order, skipping empty ones
    auto seg_count = [&](uint32_t sg) -> uint32_t {
        if (GEN) {
            const uint32_t f0 = sg * S;
            return f0 >= a.n_rays ? 0u : (a.n_rays - f0 < S ? a.n_rays - f0 : S);
        }
        return a.cnt_in[sg];
    };
    uint32_t seg = wslot, n = 0u, g = 0u;
    while (seg < a.nseg_in) {
        n = seg_count(seg);
        if (n) break;
        seg += nslots;
    }
    bool fresh_left = seg < a.nseg_in;
    // One stage ahead of a FRESH group, twelve lanes touch the 128-byte lines its six loads will read (one dword each):
    // the HBM latency passes under the stages in between and the loads themselves hit in L2.  One live VGPR.
    float warm = 0.0f;
    auto warm_up = [&]() {
        if (!GEN && fresh_left && lane < 12u) {
            const uint32_t ray = seg * S + g + (lane & 1u) * 32u;
            __builtin_assume(ray < (1u << 29));
            if (g + (lane & 1u) * 32u < n)                            // per-lane field: the plain multiply
                warm = ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, ray * 4u, (lane >> 1) * cap4, 0)) : (a.in + (size_t)(lane >> 1) * cap)[ray];
        }
    };
    warm_up();
    // output cursor
    uint32_t oseg = wslot, ofill = 0u;
    uint32_t nbox = 0u, nsph = 0u;
    const float kInf = 100000000000000000.0f;

    // One TEST group: `valid` lanes hold a queue record at `r` (per lane), all of the type `isb` says.  Pops nothing and
    // writes nothing but the accumulator: returns whether the lane's ray lives on, and that ray.
    auto test_group = [&](const bool isb, const bool valid, const float *r, f3 &o, f3 &d, f3 &thr, uint32_t &pv) -> bool {
        o = mk(0, 0, 0); d = mk(0, 0, 1); thr = mk(0, 0, 0);
        pv = 0u;
        uint32_t idx = 0u, mask = 0u;
        int j = 0;
        if (valid) {
            o = mk(r[0 * kQCap], r[1 * kQCap], r[2 * kQCap]);
            d = mk(r[3 * kQCap], r[4 * kQCap], r[5 * kQCap]);
            idx = __float_as_uint(r[6 * kQCap]);
            mask = __float_as_uint(r[7 * kQCap]);
            j = (int)__float_as_uint(r[8 * kQCap]);
            // throughput + pixel word of the ray: requested now, used after the test (nearly every tested ray is a hit)
            if (GEN) { pv = idx; thr = mk(1.0f, 1.0f, 1.0f); }
            else {
                __builtin_assume(idx < (1u << 29));
                thr = mk(ldf(6, idx), ldf(7, idx), ldf(8, idx));
                pv = __float_as_uint(ldf(9, idx));
            }
        }
        __builtin_amdgcn_wave_barrier();
        PT_MARK("exact_begin");
        // One exact test per popped ray: the queue's type for every lane (jb / jm are per lane only for meshes, which
        // share the spheres' stack).  Nearest-hit update of the reference loop: first strictly nearer wins, ties to the
        // lower index -- trivial for the first candidate.
        float best;
        int hit, face = -1;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 0);
        {
            const GeomRec *gr = lg + j;                               // per-lane gather from the LDS table
            float depth = -1.0f;
            const bool jm = MESH && ((meshbits >> j) & 1u);
            if (isb) { if (valid) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, P, face); }
            else {
                if (!MESH || __any(valid && !jm)) { if (valid && !jm) depth = sphere_test(gr->inv, gr->xf, o, d, P, N); }
                if (MESH) { if (__any(valid && jm)) { if (valid && jm) depth = mesh_test(gr, o, d, P, N); } }
            }
            const bool wins = valid && depth > -PT_EPSILON && depth < kInf;
            best = wins ? depth : kInf;
            hit = wins ? j : -1;
        }
        // The rare rays (0.04 % in the Cornell box) with rivals: candidates entered farther than the best hit are dropped,
        // the nearest of the rest is tested on the spot, until no lane has one left.
        if (__any(valid && mask != 0u)) {
            bool active = valid;
            for (;;) {
                int next_j = -1;
                if (active && mask != 0u) {
                    const CullRay cr = make_cull_ray(o, d);
                    float nt = 3.0e38f;
                    uint32_t m = mask;
                    while (m) {
                        const int jj = __builtin_ctz(m);
                        m &= m - 1u;
                        const GeomRec *gb = lg + jj;
                        float tn;
                        if ((aabbbits >> jj) & 1u) (void)cull_box(gb->bmin, gb->bmax, cr, tn);
                        else (void)cull_sphere(gb->bmin, gb->bmax, cr, tn);
                        if (hit >= 0 && tn - gb->slack > best) { mask &= ~(1u << jj); continue; }
                        if (tn < nt) { nt = tn; next_j = jj; }
                    }
                }
                active = next_j >= 0;
                if (!__any(active)) break;
                if (active) { j = next_j; mask &= ~(1u << next_j); }
                const bool jb = (boxbits >> j) & 1u;
                const bool jm = MESH && ((meshbits >> j) & 1u);
                const GeomRec *gr = lg + j;
                f3 p = mk(0, 0, 0), nn = mk(0, 0, 0);
                int fc = -1;
                float depth = -1.0f;
                if (__any(active && jb)) { if (active && jb) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, p, fc); }
                if (__any(active && !jb && !jm)) { if (active && !jb && !jm) depth = sphere_test(gr->inv, gr->xf, o, d, p, nn); }
                if (MESH) { if (__any(active && jm)) { if (active && jm) depth = mesh_test(gr, o, d, p, nn); } }
                const bool wins = active && depth > -PT_EPSILON && (depth < best || (depth == best && j < hit));
                if (wins) { best = depth; hit = j; P = p; N = nn; face = fc; }
            }
        }
        PT_MARK("exact_end");
        const bool shade = hit >= 0;
#ifdef PT_CULL_STATS
        qstat(isb ? 10 : 12, 1ull); qstat(isb ? 11 : 13, (unsigned long long)__popcll(__ballot(valid)));
        qstat(14, (unsigned long long)__popcll(__ballot(shade)));
#endif

        // -------------------------------------------------------------------- shade the hits
        PT_MARK("shade_begin");
        bool alive = false;
        if (shade) {
            const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
            const MatRec m = lm[lg[hit].mat];
            if (LAST && !(m.emittance > 0.0f)) {
                alive = true;                                         // depth exhausted: alive, contributes 0
            } else {
                const uint32_t iteration = a.iteration + slot;
                uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + (uint32_t)a.bounce));
                st = lcg_next(st); const float u_sel = u01(st);
                st = lcg_next(st); const float xi1 = u01(st);
                st = lcg_next(st); const float xi2 = u01(st);
                f3 L = mk(0.0f, 0.0f, 0.0f);
                int code = 4;
                const bool hb = (boxbits >> hit) & 1u;
                if (__any(hb)) { if (hb) code = scatter_box(m, P, face, lf + 3 * hit, u_sel, xi1, xi2, o, d, thr, L); }
                if (__any(!hb)) { if (!hb) code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, L); }
                if (code == 3) {
#if PT_Q_PARK
                    float *base = reinterpret_cast<float *>((uintptr_t)((unsigned long long)park[0] | ((unsigned long long)park[1] << 32)));
                    size_t off = (size_t)pixel * 3;
                    if (a.batch > 1u) {
                        const uint32_t W = park[4];
                        const uint32_t y = (uint32_t)(((unsigned long long)pixel * park[6]) >> park[7]);
                        const uint32_t x = pixel - y * W;
                        const uint32_t ly = (uint32_t)(((unsigned long long)(y - park[5]) * park[8]) >> park[9]);
                        off = (size_t)slot * (size_t)((unsigned long long)park[2] | ((unsigned long long)park[3] << 32)) + (size_t)(ly * W + x) * 3;
                    }
                    float *px = base + off;
#else
                    float *px = a.batch > 1u ? a.planes + (size_t)slot * a.plane_stride + (size_t)owned_index(a.cam, pixel) * 3
                                             : a.image + (size_t)pixel * 3;
#endif
                    (void)unsafeAtomicAdd(px, L.x); (void)unsafeAtomicAdd(px + 1, L.y); (void)unsafeAtomicAdd(px + 2, L.z);
                    emitted++;
                }
                alive = code <= 2;
            }
        }
        return alive;
    };

    for (;;) {
        int act;
        if (nbox >= 64u) act = 1;
        else if (nsph >= 64u) act = 2;
        else if (fresh_left && nbox + nsph <= kQCap - 64u) act = 0;
#if PT_Q_MERGED_DRAIN
        else if (!fresh_left) break;                      // input exhausted, both stacks below a full group: the block's merged drain
#else
        else if (nbox + nsph == 0u) break;
#endif
        else act = nbox >= nsph ? 1 : 2;                  // no room for a fresh group: the fuller stack pops what it has

        if (act == 0) {
            // ---------------------------------------------------------------- FRESH
            PT_MARK("fresh_begin");
            __asm__ volatile("" :: "v"(warm));                        // the warm-up load is complete before its line is re-read
            const uint32_t k = g + lane;
            const bool valid = k < n;
            const uint32_t ray = seg * S + k;
            f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            uint32_t idx = 0u;
            if (valid) {
                if (GEN) {
                    const uint32_t slot = a.batch > 1u ? ray / a.n_own : 0u;
                    const uint32_t local = ray - slot * a.n_own;
                    const uint32_t W = (uint32_t)a.cam.W;
                    const uint32_t lr = local / W, x = local - lr * W;
                    const uint32_t pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
                    camera_ray(a.cam, pixel, a.iteration + slot, o, d);
                    idx = pixel | (slot << 24);                       // bounce 0 carries the pixel word itself
                } else {
                    __builtin_assume(ray < (1u << 29));
                    o = mk(ldf(0, ray), ldf(1, ray), ldf(2, ray));
                    d = mk(ldf(3, ray), ldf(4, ray), ldf(5, ray));
                    idx = ray;
                }
            }
            g += 64u;
            if (g >= n) {                                             // next non-empty segment of this wave
                g = 0u; n = 0u;
                seg += nslots;
                while (seg < a.nseg_in) {
                    n = seg_count(seg);
                    if (n) break;
                    seg += nslots;
                }
                fresh_left = seg < a.nseg_in;
            }
            warm_up();
            PT_MARK("cull_begin");
            // conservative candidate mask + nearest candidate; wave-uniform table index
            const CullRay cr = make_cull_ray(o, d);
            float near_t = 3.0e38f;
            uint32_t mask = 0u, next_j = 0u;
#pragma unroll 2
            for (int i = 0; i < qt.nbox; ++i) {
                const CullRec r = lc[i];
                float tn;
                const bool keep = cull_box(r.a, r.b, cr, tn);
                mask |= keep ? __float_as_uint(r.b[3]) : 0u;
                const bool nearer = keep && tn < near_t;
                near_t = nearer ? tn : near_t;
                next_j = nearer ? __float_as_uint(r.a[3]) : next_j;
            }
#pragma unroll 2
            for (int i = qt.nbox; i < qt.nbox + qt.nsph; ++i) {
                const CullRec r = lc[i];
                float tn;
                const bool keep = cull_sphere(r.a, r.b, cr, tn);       // reads a[0..3] and b[3], like bmin / bmax of the full record
                mask |= keep ? __float_as_uint(r.b[1]) : 0u;
                const bool nearer = keep && tn < near_t;
                near_t = nearer ? tn : near_t;
                next_j = nearer ? __float_as_uint(r.b[0]) : next_j;
            }
            PT_MARK("cull_end");
            if (!valid) mask = 0u;
            const bool push = mask != 0u;
#ifdef PT_CULL_STATS
            qstat(8, 1ull); qstat(9, (unsigned long long)__popcll(__ballot(valid)));
            atomicAdd(&g_cull_stats[5], (unsigned long long)__popc(mask));
#endif
            mask &= ~(1u << next_j);
            const bool tobox = push && ((boxbits >> next_j) & 1u);
            const u64 bb = __ballot(tobox), sb = __ballot(push && !tobox);
            if (bb | sb) {
                if (push) {
                    const uint32_t pos = tobox ? nbox + wave_rank(bb) : kQCap - 1u - (nsph + wave_rank(sb));
                    float *r = q + pos;
                    r[0 * kQCap] = o.x; r[1 * kQCap] = o.y; r[2 * kQCap] = o.z;
                    r[3 * kQCap] = d.x; r[4 * kQCap] = d.y; r[5 * kQCap] = d.z;
                    r[6 * kQCap] = __uint_as_float(idx);
                    r[7 * kQCap] = __uint_as_float(mask);
                    r[8 * kQCap] = __uint_as_float(next_j);
                }
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            PT_MARK("fresh_end");
            continue;
        }

        // -------------------------------------------------------------------- TEST (one type per group)
        PT_MARK("test_begin");
        const bool isb = act == 1;
        const uint32_t have = isb ? nbox : nsph;
        const uint32_t cnt = have < 64u ? have : 64u;
        const bool valid = lane < cnt;
        const uint32_t pos = isb ? (have - cnt + lane) : (kQCap - 1u - (have - cnt + lane));
        if (isb) nbox -= cnt; else nsph -= cnt;
        f3 o, d, thr;
        uint32_t pv;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool alive = test_group(isb, valid, q + pos, o, d, thr, pv);

        // -------------------------------------------------------------------- survivors -> the wave's output stream
        PT_MARK("out_begin");
        const u64 ballot = __ballot(alive);
        const uint32_t na = (uint32_t)__popcll(ballot);
        if (!LAST && na) {
            if (alive) {
                uint32_t p = ofill + wave_rank(ballot);
                const uint32_t sg = p >= S ? oseg + nslots : oseg;
                p = p >= S ? p - S : p;
                const uint32_t oi = sg * S + p;
                __builtin_assume(oi < (1u << 29));
                stf(0, oi, o.x); stf(1, oi, o.y); stf(2, oi, o.z);
                stf(3, oi, d.x); stf(4, oi, d.y); stf(5, oi, d.z);
                stf(6, oi, thr.x); stf(7, oi, thr.y); stf(8, oi, thr.z);
                stf(9, oi, __uint_as_float(pv));
            }
            ofill += na;
            if (ofill >= S) {
                if (lane == 0) a.cnt_out[oseg] = S;
                oseg += nslots;
                ofill -= S;
            }
        }
        survivors += na;
        PT_MARK("loop_end");
    }
#if PT_Q_MERGED_DRAIN
    // -------------------------------------------------------------------- the block's merged drain
    // Every wave arrives here once, with fewer than 64 records on either stack.  The block's leftovers of one type, taken
    // in wave order, form ceil(total / 64) groups instead of one partly filled group per wave; the groups go round the
    // waves.  A survivor is appended to the output stream of the wave that QUEUED the ray (a stream holds exactly the
    // survivors of its own wave's input, so it can never outgrow its segments): cursors in LDS, advanced by ds_add.
    {
        uint32_t *ep = ctrl + 2;                              // [0..3] box counts, [4..7] sphere counts, [8..11] stream fill, [12..15] stream segment
        if (lane == 0) { ep[wave] = nbox; ep[4u + wave] = nsph; ep[8u + wave] = ofill; ep[12u + wave] = oseg; }
        __syncthreads();
        uint32_t cb[kWaves], cs[kWaves], dseg[kWaves];
        uint32_t tb = 0u, ts = 0u;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)kWaves; ++w) {
            cb[w] = __builtin_amdgcn_readfirstlane(ep[w]); cs[w] = __builtin_amdgcn_readfirstlane(ep[4u + w]);
            dseg[w] = __builtin_amdgcn_readfirstlane(ep[12u + w]);
            tb += cb[w]; ts += cs[w];
        }
        const uint32_t gb = (tb + 63u) >> 6, gs = (ts + 63u) >> 6;
        const float *qall = reinterpret_cast<const float *>(smem + q_lds_offset(a.G, a.M));
        for (uint32_t grp = wave; grp < gb + gs; grp += (uint32_t)kWaves) {
            const bool isb = grp < gb;
            const uint32_t i = ((isb ? grp : grp - gb) << 6) + lane;
            const bool valid = i < (isb ? tb : ts);
            // the wave that queued record i, and the record's place on that wave's stack
            uint32_t donor = 0u, before = 0u, run = 0u;
#pragma unroll
            for (uint32_t w = 0; w + 1u < (uint32_t)kWaves; ++w) {
                run += isb ? cb[w] : cs[w];
                if (i >= run) { donor = w + 1u; before = run; }
            }
            const uint32_t pos = isb ? (i - before) : (kQCap - 1u - (i - before));
            f3 o, d, thr;
            uint32_t pv;
            const bool alive = test_group(isb, valid, qall + (size_t)donor * kQCap * kQFields + pos, o, d, thr, pv);
            const u64 ballot = __ballot(alive);
            if (!LAST && ballot) {
                uint32_t p = 0u, sg = 0u;
#pragma unroll
                for (uint32_t w = 0; w < (uint32_t)kWaves; ++w) {
                    const u64 bw = __ballot(alive && donor == w);
                    if (bw) {
                        uint32_t base = 0u;
                        if (lane == (uint32_t)__builtin_ctzll(bw)) base = atomicAdd(&ep[8u + w], (uint32_t)__popcll(bw));
                        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(bw));
                        if (alive && donor == w) { p = base + wave_rank(bw); sg = dseg[w]; }
                    }
                }
                if (alive) {
                    if (p >= S) { p -= S; sg += nslots; }
                    const uint32_t oi = sg * S + p;
                    __builtin_assume(oi < (1u << 29));
                    stf(0, oi, o.x); stf(1, oi, o.y); stf(2, oi, o.z);
                    stf(3, oi, d.x); stf(4, oi, d.y); stf(5, oi, d.z);
                    stf(6, oi, thr.x); stf(7, oi, thr.y); stf(8, oi, thr.z);
                    stf(9, oi, __uint_as_float(pv));
                }
            }
            survivors += (uint32_t)__popcll(ballot);
        }
        if (!LAST) {
            __syncthreads();                                  // every group of the block has advanced the cursors
            ofill = __builtin_amdgcn_readfirstlane(ep[8u + wave]);
            if (ofill >= S) {                                 // at most one segment boundary: a block's leftovers are below 4 x 126 < 3 S ... per stream below 126 < S
                if (lane == 0) a.cnt_out[oseg] = S;
                oseg += nslots;
                ofill -= S;
            }
        }
    }
#endif
    // close the output stream: the partly filled segment, then zeros for the wave's unused ones
    if (!LAST && lane == 0) {
        uint32_t sg = oseg, c = ofill;
        while (sg < a.nseg_out) { a.cnt_out[sg] = c; c = 0u; sg += nslots; }
    }

    for (int sft = 32; sft > 0; sft >>= 1) emitted += __shfl_down(emitted, sft);
    if (lane == 0) {
        if (survivors) atomicAdd(&ctrl[0], survivors);
        if (emitted) atomicAdd(&ctrl[1], emitted);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ctrl[0]) atomicAdd(&bank[a.bounce + 1], ctrl[0]);
        if (ctrl[1]) atomicAdd(&a.sync->emitted, (u64)ctrl[1]);
    }
}

// ------------------------------------------------------------------ whole paths on typed work queues ---
// ordering = 2: ONE launch per group takes every ray from the camera to its end.  The two stages are those of
// k_bounce_q, but nothing is per bounce any more:
//   * a wave draws small jobs of camera rays from a device ticket counter (dynamic balance, one short tail per launch);
//   * a queue record is the whole ray (origin, direction, throughput, pixel word) + candidate mask, nearest candidate and
//     its LEVEL (bounce index), so records of different bounces share the two typed stacks and a partly filled stack
//     never has to be popped before the launch ends;
//   * survivors of a TEST group go on the wave's own small STACK of rays in global memory (256 slots x 44 B: the ray and
//     its level), written and read back by the same wave within a few groups -- L2 traffic, 70 MB for the whole chip
//     instead of two 0.8-GB pools; FRESH takes the top 64 rays of the stack when it holds 64 (depth first: the stack never
//     holds more than 63 + 2 x 64 rays), else a group of camera rays, else whatever is left.
// No barrier, no inter-wave traffic, no pool; per-level live counts through one LDS atomic per group.  Results: the same image, live
// counts and emitter hits as every other kernel (the rays of a bounce are a set, not a sequence).
#ifndef PT_P_CAP
#define PT_P_CAP 138                     // records per wave (12 dwords each): 5 blocks = 20 waves per CU; 4 blocks with more records and 6 with fewer measured slower
#endif
constexpr uint32_t kPCap = PT_P_CAP;
constexpr uint32_t kPFields = 12;        // ox oy oz dx dy dz tx ty tz pixelword mask candidate|level<<8
constexpr uint32_t kStack = 256;         // rays on a wave's stack (bound: 63 + two pops of 64)
constexpr uint32_t kSFields = 11;        // ox oy oz dx dy dz tx ty tz pixelword level
constexpr uint32_t kTicketCtrs = 16, kTicketStride = 64;
constexpr uint32_t kJobMax = 128;        // camera rays per job: about 1/48 of a wave's share of the launch, 64 .. kJobMax

struct PathArgs {
    float *arena;                    // [waves][kSFields][kStack]
    uint32_t arena_bytes;            // != 0: below 4 GiB, buffer addressing
    uint32_t depth;
    uint32_t *ticket;                // kTicketCtrs counters, kTicketStride dwords apart (one cache line each), zero before the launch:
                                     //   counter k hands out the drawn jobs k, k + kTicketCtrs, ... -- same-address device atomics are served
                                     //   one after the other (~10 ns each), sixteen lines sixteen times as fast
    uint32_t job_rays;               // camera rays per job (a multiple of 64)
    uint32_t static_rounds;          // every wave's first jobs are its own (job = round * waves + slot): device atomics on ONE
                                     //   address are served memory-side at ~8 ns each, so only the last part of a launch is drawn
    uint32_t *error;
};

__host__ __device__ inline uint32_t p_cursor_offset(int G, int M) { return q_lds_offset(G, M); }
__host__ __device__ inline uint32_t p_queue_offset(int G, int M) { return p_cursor_offset(G, M); }
__host__ __device__ inline uint32_t p_lds_bytes(int G, int M) { return p_queue_offset(G, M) + (uint32_t)kWaves * kPCap * kPFields * 4u; }

// MESH: the scene holds MESH primitives with triangles (they share the spheres' stack; the traversal needs registers the
// common variant must not pay for)
template <bool MESH>
__global__ __launch_bounds__(kBlock, 4) void k_path_q(SegArgs a, PathArgs pa, const GeomRec *__restrict__ geoms,
                                                      const MatRec *__restrict__ mats, QTables qt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);   // [1] emitted (block sum), [32..96] survivors per level
    uint32_t *lsurv = ctrl + 32;
    if (threadIdx.x < 2) ctrl[threadIdx.x] = 0u;
    if (threadIdx.x < 65) lsurv[threadIdx.x] = 0u;
    uint32_t *park = ctrl + 18;
    if (threadIdx.x == 0) {
        const unsigned long long pl = (unsigned long long)(uintptr_t)(a.batch > 1u ? a.planes : a.image), st = (unsigned long long)a.plane_stride;
        park[0] = (uint32_t)pl; park[1] = (uint32_t)(pl >> 32); park[2] = (uint32_t)st; park[3] = (uint32_t)(st >> 32);
        park[4] = (uint32_t)a.cam.W; park[5] = (uint32_t)a.cam.row_offset; park[6] = a.cam.mW; park[7] = a.cam.shW;
        park[8] = a.cam.mS; park[9] = a.cam.shS;
    }
    GeomRec *lg;
    MatRec *lm;
    FaceFrame *lf = reinterpret_cast<FaceFrame *>(smem + q_frames_offset(a.G, a.M));
    CullRec *lc = reinterpret_cast<CullRec *>(smem + q_cull_offset(a.G, a.M));
    {
        uint32_t *fd = reinterpret_cast<uint32_t *>(lf);
        const uint32_t *fs = reinterpret_cast<const uint32_t *>(qt.frames);
        for (uint32_t i = threadIdx.x; i < (uint32_t)a.G * 3u * (uint32_t)(sizeof(FaceFrame) / 4); i += blockDim.x) fd[i] = fs[i];
        uint32_t *cd = reinterpret_cast<uint32_t *>(lc);
        const uint32_t *cs = reinterpret_cast<const uint32_t *>(qt.cull);
        for (uint32_t i = threadIdx.x; i < (uint32_t)(qt.nbox + qt.nsph) * (uint32_t)(sizeof(CullRec) / 4); i += blockDim.x) cd[i] = cs[i];
    }
    stage_tables(smem, geoms, a.G, mats, a.M, true, lg, lm);        // ends with __syncthreads()

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t wslot = blockIdx.x * kWaves + wave;
    const uint32_t D = pa.depth;
    float *q = reinterpret_cast<float *>(smem + p_queue_offset(a.G, a.M)) + (size_t)wave * kPCap * kPFields;
    // the wave's stack: field f of slot s at woff + f * kStack + s
    const uint32_t wave_floats = kSFields * kStack;
    const bool ub = pa.arena_bytes != 0u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.arena, 0, pa.arena_bytes, 0x00020000);
    const uint32_t woff = wslot * wave_floats;             // in floats (buffer path: arena below 4 GiB)
    auto ring_ld = [&](uint32_t off, uint32_t f) -> float {         // off = float index of field 0 of the slot
        return ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, f * kStack * 4u, 0)) : pa.arena[(size_t)off + f * kStack];
    };
    auto ring_st = [&](uint32_t off, uint32_t f, float v) {
        if (ub) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off * 4u, f * kStack * 4u, 0);
        else pa.arena[(size_t)off + f * kStack] = v;
    };

    uint32_t *bank = a.bank ? a.sync->counts_b : a.sync->counts;
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        uint32_t *other = a.bank ? a.sync->counts : a.sync->counts_b;
        a.sync->totals[threadIdx.x] += other[threadIdx.x];
        other[threadIdx.x] = 0u;
        if (threadIdx.x == 0) bank[0] = a.n_rays;
    }
    uint32_t boxbits = 0u, meshbits = 0u;
    for (int j = 0; j < a.G; ++j) {
        if (lg[j].type == 1) boxbits |= 1u << j;
        else if (MESH && lg[j].type == 2 && lg[j].inside_hits != 0) meshbits |= 1u << j;
    }
    const uint32_t aabbbits = boxbits | meshbits;         // primitives whose conservative bound is a box

    uint32_t emitted = 0u;
    uint32_t nbox = 0u, nsph = 0u;
    uint32_t sp = 0u;                                      // rays on the wave's stack
    uint32_t jobpos = 0u, jobend = 0u;
    bool tickets_left = true;
    // the ticket of the NEXT job is requested one job ahead (lane 0 holds it): its latency passes under the current job
    uint32_t next_ticket = 0u;
    uint32_t round = 0u;                                   // static jobs taken so far
    const uint32_t nwaves = gridDim.x * kWaves;
    uint32_t ctr = wslot % kTicketCtrs, dry = 0u;          // the counter this wave draws from; counters found exhausted in a row
    if (pa.static_rounds == 0u && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);
    uint32_t turns = 0u;
    const float kInf = 100000000000000000.0f;

    for (;;) {
        if (++turns > (1u << 24)) { if (lane == 0) *pa.error = 3u; break; }                // never reached; bounds a broken build
        int act;
        if (nbox >= 64u) act = 1;
        else if (nsph >= 64u) act = 2;
        else if (nbox + nsph <= kPCap - 64u) {
            if (sp >= 64u) act = 0;
            else {
                while (jobpos >= jobend && tickets_left) {                                 // next job of camera rays (a dry counter: try the next)
                    unsigned long long job;
                    if (round < pa.static_rounds) {
                        job = (unsigned long long)wslot * pa.static_rounds + round;       // the wave's own contiguous range: neighbouring rays take similar paths
                        round++;
                        if (round == pa.static_rounds && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);      // first drawn job, one job ahead
                    } else {
                        job = (unsigned long long)pa.static_rounds * nwaves + (unsigned long long)__builtin_amdgcn_readfirstlane(next_ticket) * kTicketCtrs + ctr;
                        if (job * pa.job_rays >= (unsigned long long)a.n_rays) {          // this counter is dry: on to the next one
                            dry++;
                            ctr = ctr + 1u == kTicketCtrs ? 0u : ctr + 1u;
                            if (dry >= kTicketCtrs) tickets_left = false;
                        } else dry = 0u;
                        if (tickets_left && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);
                    }
                    const unsigned long long first = job * pa.job_rays;
                    if (first < (unsigned long long)a.n_rays) { jobpos = (uint32_t)first; jobend = a.n_rays - jobpos < pa.job_rays ? a.n_rays : jobpos + pa.job_rays; }
                }
                if (jobpos < jobend) act = 3;
                else if (sp) act = 0;
                else if (nbox + nsph) act = nbox >= nsph ? 1 : 2;
                else break;
            }
        } else act = nbox >= nsph ? 1 : 2;

        if (act == 0 || act == 3) {
            // ---------------------------------------------------------------- FRESH
            f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1.0f, 1.0f, 1.0f);
            uint32_t pv = 0u, level = 0u;
            bool valid;
            if (act == 3) {                                                                // camera rays
                const uint32_t ray = jobpos + lane;
                valid = ray < jobend;
                jobpos = jobpos + 64u < jobend ? jobpos + 64u : jobend;
                if (valid) {
                    const uint32_t slot = a.batch > 1u ? ray / a.n_own : 0u;
                    const uint32_t local = ray - slot * a.n_own;
                    const uint32_t W = (uint32_t)a.cam.W;
                    const uint32_t lr = local / W, x = local - lr * W;
                    const uint32_t pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
                    camera_ray(a.cam, pixel, a.iteration + slot, o, d);
                    pv = pixel | (slot << 24);
                }
            } else {                                                                       // the top of the wave's stack
                const uint32_t cnt = sp < 64u ? sp : 64u;
                valid = lane < cnt;
                sp -= cnt;
                const uint32_t off = woff + sp + lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the wave's own stack stores have landed (vmcnt 0) ...
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");        // ... before they are read back through the same L1
                if (valid) {
                    o = mk(ring_ld(off, 0), ring_ld(off, 1), ring_ld(off, 2));
                    d = mk(ring_ld(off, 3), ring_ld(off, 4), ring_ld(off, 5));
                    thr = mk(ring_ld(off, 6), ring_ld(off, 7), ring_ld(off, 8));
                    pv = __float_as_uint(ring_ld(off, 9));
                    level = __float_as_uint(ring_ld(off, 10));
                }
            }
            const CullRay cr = make_cull_ray(o, d);
            float near_t = 3.0e38f;
            uint32_t mask = 0u, next_j = 0u;
#pragma unroll 2
            for (int i = 0; i < qt.nbox; ++i) {
                const CullRec r = lc[i];
                float tn;
                const bool keep = cull_box(r.a, r.b, cr, tn);
                mask |= keep ? __float_as_uint(r.b[3]) : 0u;
                const bool nearer = keep && tn < near_t;
                near_t = nearer ? tn : near_t;
                next_j = nearer ? __float_as_uint(r.a[3]) : next_j;
            }
#pragma unroll 2
            for (int i = qt.nbox; i < qt.nbox + qt.nsph; ++i) {
                const CullRec r = lc[i];
                float tn;
                const bool keep = cull_sphere(r.a, r.b, cr, tn);
                mask |= keep ? __float_as_uint(r.b[1]) : 0u;
                const bool nearer = keep && tn < near_t;
                near_t = nearer ? tn : near_t;
                next_j = nearer ? __float_as_uint(r.b[0]) : next_j;
            }
            if (!valid) mask = 0u;
            const bool push = mask != 0u;
#ifdef PT_CULL_STATS
            qstat(8, 1ull); qstat(9, (unsigned long long)__popcll(__ballot(valid)));
            atomicAdd(&g_cull_stats[5], (unsigned long long)__popc(mask));
            if (act == 3) qstat(6, 1ull);
#endif
            mask &= ~(1u << next_j);
            const bool tobox = push && ((boxbits >> next_j) & 1u);
            const u64 bb = __ballot(tobox), sb = __ballot(push && !tobox);
            if (bb | sb) {
                if (push) {
                    const uint32_t pos = tobox ? nbox + wave_rank(bb) : kPCap - 1u - (nsph + wave_rank(sb));
                    float *r = q + pos;
                    r[0 * kPCap] = o.x; r[1 * kPCap] = o.y; r[2 * kPCap] = o.z;
                    r[3 * kPCap] = d.x; r[4 * kPCap] = d.y; r[5 * kPCap] = d.z;
                    r[6 * kPCap] = thr.x; r[7 * kPCap] = thr.y; r[8 * kPCap] = thr.z;
                    r[9 * kPCap] = __uint_as_float(pv);
                    r[10 * kPCap] = __uint_as_float(mask);
                    r[11 * kPCap] = __uint_as_float(next_j | (level << 8));
                }
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }

        // -------------------------------------------------------------------- TEST (one type per group, any levels)
        const bool isb = act == 1;
        const uint32_t have = isb ? nbox : nsph;
        const uint32_t cnt = have < 64u ? have : 64u;
        const bool valid = lane < cnt;
        const uint32_t pos = isb ? (have - cnt + lane) : (kPCap - 1u - (have - cnt + lane));
        if (isb) nbox -= cnt; else nsph -= cnt;
#ifdef PT_CULL_STATS
        qstat(isb ? 10 : 12, 1ull); qstat(isb ? 11 : 13, (unsigned long long)cnt);
#endif
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(0, 0, 0);
        uint32_t pv = 0u, mask = 0u, level = 0u;
        int j = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            const float *r = q + pos;
            o = mk(r[0 * kPCap], r[1 * kPCap], r[2 * kPCap]);
            d = mk(r[3 * kPCap], r[4 * kPCap], r[5 * kPCap]);
            thr = mk(r[6 * kPCap], r[7 * kPCap], r[8 * kPCap]);
            pv = __float_as_uint(r[9 * kPCap]);
            mask = __float_as_uint(r[10 * kPCap]);
            const uint32_t jl = __float_as_uint(r[11 * kPCap]);
            j = (int)(jl & 0xFFu);
            level = jl >> 8;
        }
        __builtin_amdgcn_wave_barrier();
        float best;
        int hit, face = -1;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 0);
        {
            const GeomRec *gr = lg + j;
            float depth = -1.0f;
            const bool jm = MESH && ((meshbits >> j) & 1u);
            if (isb) { if (valid) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, P, face); }
            else {
                if (!MESH || __any(valid && !jm)) { if (valid && !jm) depth = sphere_test(gr->inv, gr->xf, o, d, P, N); }
                if (MESH) { if (__any(valid && jm)) { if (valid && jm) depth = mesh_test(gr, o, d, P, N); } }
            }
            const bool wins = valid && depth > -PT_EPSILON && depth < kInf;
            best = wins ? depth : kInf;
            hit = wins ? j : -1;
        }
        if (__any(valid && mask != 0u)) {
            bool active = valid;
            for (;;) {
                int next_j = -1;
                if (active && mask != 0u) {
                    const CullRay cr = make_cull_ray(o, d);
                    float nt = 3.0e38f;
                    uint32_t m = mask;
                    while (m) {
                        const int jj = __builtin_ctz(m);
                        m &= m - 1u;
                        const GeomRec *gb = lg + jj;
                        float tn;
                        if ((aabbbits >> jj) & 1u) (void)cull_box(gb->bmin, gb->bmax, cr, tn);
                        else (void)cull_sphere(gb->bmin, gb->bmax, cr, tn);
                        if (hit >= 0 && tn - gb->slack > best) { mask &= ~(1u << jj); continue; }
                        if (tn < nt) { nt = tn; next_j = jj; }
                    }
                }
                active = next_j >= 0;
                if (!__any(active)) break;
                if (active) { j = next_j; mask &= ~(1u << next_j); }
                const bool jb = (boxbits >> j) & 1u;
                const bool jm = MESH && ((meshbits >> j) & 1u);
                const GeomRec *gr = lg + j;
                f3 p = mk(0, 0, 0), nn = mk(0, 0, 0);
                int fc = -1;
                float depth = -1.0f;
                if (__any(active && jb)) { if (active && jb) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, p, fc); }
                if (__any(active && !jb && !jm)) { if (active && !jb && !jm) depth = sphere_test(gr->inv, gr->xf, o, d, p, nn); }
                if (MESH) { if (__any(active && jm)) { if (active && jm) depth = mesh_test(gr, o, d, p, nn); } }
                const bool wins = active && depth > -PT_EPSILON && (depth < best || (depth == best && j < hit));
                if (wins) { best = depth; hit = j; P = p; N = nn; face = fc; }
            }
        }
#ifdef PT_CULL_STATS
        qstat(14, (unsigned long long)__popcll(__ballot(hit >= 0)));
        qstat(7, (unsigned long long)__popcll(__ballot(hit >= 0 && level + 1u >= D)));
#endif
        // shade the hits (the ray's own level is its bounce index)
        bool alive = false;
        if (hit >= 0) {
            const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
            const MatRec m = lm[lg[hit].mat];
            if (level + 1u >= D && !(m.emittance > 0.0f)) {
                alive = true;                                         // depth exhausted: alive, contributes 0
            } else {
                const uint32_t iteration = a.iteration + slot;
                uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + level));
                st = lcg_next(st); const float u_sel = u01(st);
                st = lcg_next(st); const float xi1 = u01(st);
                st = lcg_next(st); const float xi2 = u01(st);
                f3 L = mk(0.0f, 0.0f, 0.0f);
                int code = 4;
                const bool hb = (boxbits >> hit) & 1u;
                if (__any(hb)) { if (hb) code = scatter_box(m, P, face, lf + 3 * hit, u_sel, xi1, xi2, o, d, thr, L); }
                if (__any(!hb)) { if (!hb) code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, L); }
                if (code == 3) {
                    float *base = reinterpret_cast<float *>((uintptr_t)((unsigned long long)park[0] | ((unsigned long long)park[1] << 32)));
                    size_t off = (size_t)pixel * 3;
                    if (a.batch > 1u) {
                        const uint32_t W = park[4];
                        const uint32_t y = (uint32_t)(((unsigned long long)pixel * park[6]) >> park[7]);
                        const uint32_t x = pixel - y * W;
                        const uint32_t ly = (uint32_t)(((unsigned long long)(y - park[5]) * park[8]) >> park[9]);
                        off = (size_t)slot * (size_t)((unsigned long long)park[2] | ((unsigned long long)park[3] << 32)) + (size_t)(ly * W + x) * 3;
                    }
                    float *px = base + off;
                    (void)unsafeAtomicAdd(px, L.x); (void)unsafeAtomicAdd(px + 1, L.y); (void)unsafeAtomicAdd(px + 2, L.z);
                    emitted++;
                }
                alive = code <= 2;
            }
        }
        // survivors: counted per level (one LDS atomic for the group); those with bounces left go on the wave's stack
        if (alive) atomicAdd(&lsurv[level + 1u], 1u);
        const bool onward = alive && level + 1u < D;
        const u64 ballot = __ballot(onward);
        if (ballot) {
            const uint32_t n = (uint32_t)__popcll(ballot);
            if (sp + n > kStack) { if (lane == 0) *pa.error = 2u; }        // never: see the bound above
            else {
                if (onward) {
                    const uint32_t off = woff + sp + wave_rank(ballot);
                    ring_st(off, 0, o.x); ring_st(off, 1, o.y); ring_st(off, 2, o.z);
                    ring_st(off, 3, d.x); ring_st(off, 4, d.y); ring_st(off, 5, d.z);
                    ring_st(off, 6, thr.x); ring_st(off, 7, thr.y); ring_st(off, 8, thr.z);
                    ring_st(off, 9, __uint_as_float(pv));
                    ring_st(off, 10, __uint_as_float(level + 1u));
                }
                sp += n;
            }
        }
    }

    for (int sft = 32; sft > 0; sft >>= 1) emitted += __shfl_down(emitted, sft);
    if (lane == 0 && emitted) atomicAdd(&ctrl[1], emitted);
    __syncthreads();
    if (threadIdx.x == 0 && ctrl[1]) atomicAdd(&a.sync->emitted, (u64)ctrl[1]);
    if (threadIdx.x >= 1 && threadIdx.x < 65 && lsurv[threadIdx.x]) atomicAdd(&bank[threadIdx.x], lsurv[threadIdx.x]);
}

// ------------------------------------------------------------------ fold (batched iterations) ---
// image[p] = (((image[p] + plane_0[p]) + plane_1[p]) + ...) in iteration order -- the same sum, in the
// same order, as rendering the iterations one after the other -- and clears the planes for the next
// batch.  One thread per owned pixel.
struct FoldArgs {
    float *image;
    float *planes;
    size_t plane_stride;
    uint32_t batch, n_own;
    int W, row_offset, row_stride;
};

__global__ __launch_bounds__(kBlock) void k_fold(FoldArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.n_own) return;
    const uint32_t W = (uint32_t)a.W;
    const uint32_t lr = gid / W, x = gid - lr * W;
    const size_t p = ((size_t)(lr * (uint32_t)a.row_stride + (uint32_t)a.row_offset) * W + x) * 3;
    float r = a.image[p], g = a.image[p + 1], b = a.image[p + 2];
    for (uint32_t s = 0; s < a.batch; ++s) {
        float *q = a.planes + (size_t)s * a.plane_stride + (size_t)gid * 3;       // planes hold the owned pixels only
        r = r + q[0]; g = g + q[1]; b = b + q[2];
        q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
    }
    a.image[p] = r; a.image[p + 1] = g; a.image[p + 2] = b;
}

// ------------------------------------------------------------------ flat (reference) ---
struct FlatArgs {
    CamRec cam;
    float *image;
    const GeomRec *geoms;
    const MatRec *mats;
    int G, M;
    uint32_t n_own;
    // optional debug outputs
    float *dir, *t, *P, *N;
    int *hit;
    int write_image;
};

// raytraceRay as shipped (src/raytraceKernel.cu:123-159): nearest hit, flat material colour
// OVERWRITES the pixel, misses leave it untouched.  Also the primary-hit parity hook.
__global__ __launch_bounds__(kBlock) void k_flat(FlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GeomRec *lg;
    MatRec *lm;
    stage_tables(smem, a.geoms, a.G, a.mats, a.M, true, lg, lm);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.n_own) return;
    const uint32_t W = (uint32_t)a.cam.W;
    const uint32_t lr = gid / W, x = gid - lr * W;
    const uint32_t pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
    f3 o, d, P = mk(0, 0, 0), N = mk(0, 0, 0);
    CamRec c = a.cam;
    c.camera_mode = 0; c.antialias = 0;
    camera_ray(c, pixel, 1u, o, d);
    float t;
    const int hit = nearest_hit(lg, a.G, o, d, t, P, N);
    if (a.write_image && hit >= 0) {
        const MatRec m = lm[lg[hit].mat];
        float *px = a.image + (size_t)pixel * 3;
        px[0] = m.color[0]; px[1] = m.color[1]; px[2] = m.color[2];
    }
    if (a.hit) a.hit[pixel] = hit;
    if (a.t) a.t[pixel] = t;
    if (a.dir) { a.dir[3 * pixel] = d.x; a.dir[3 * pixel + 1] = d.y; a.dir[3 * pixel + 2] = d.z; }
    if (a.P) { a.P[3 * pixel] = P.x; a.P[3 * pixel + 1] = P.y; a.P[3 * pixel + 2] = P.z; }
    if (a.N) { a.N[3 * pixel] = N.x; a.N[3 * pixel + 1] = N.y; a.N[3 * pixel + 2] = N.z; }
}

// sendImageToPBO (src/raytraceKernel.cu:88-119); scale = 1 is the reference
__global__ __launch_bounds__(kBlock) void k_display(const float *image, uchar4 *out, uint32_t n, float scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = (image[3 * i] * scale) * 255.0f, g = (image[3 * i + 1] * scale) * 255.0f,
          b = (image[3 * i + 2] * scale) * 255.0f;
    if (r > 255.0f) r = 255.0f;
    if (g > 255.0f) g = 255.0f;
    if (b > 255.0f) b = 255.0f;
    uchar4 v;
    v.w = 0; v.x = (unsigned char)r; v.y = (unsigned char)g; v.z = (unsigned char)b;
    out[i] = v;
}

// ------------------------------------------------------------------ KAT kernels --------
// generateRandomNumberFromThread (src/raytraceKernel.cu:30-37)
__global__ void k_rng_from_thread(float resx, float time, int n, const int *xy, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    const int index = (int)((float)x + ((float)y * resx));
    const uint32_t s = (uint32_t)((float)index * time);
    uint32_t st = lcg_seed(hash(s));
    st = lcg_next(st); out[3 * i] = u01(st);
    st = lcg_next(st); out[3 * i + 1] = u01(st);
    st = lcg_next(st); out[3 * i + 2] = u01(st);
}

__global__ void k_hemisphere(int n, const float *nrm, const float *xi, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 r = hemisphere(mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), xi[2 * i], xi[2 * i + 1]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

__global__ void k_light_points(const GeomRec *g, int n, const float *seeds, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 p = g->type == 0 ? random_point_on_sphere(g->xf, seeds[i]) : random_point_on_cube(g->xf, seeds[i]);
    out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
}

__global__ void k_sincos(int n, const float *a, float *s, float *c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincos_poly(a[i], sn, cs);
    s[i] = sn; c[i] = cs;
}

}  // namespace

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
    u64 *d_status = nullptr;
    uint32_t max_chunks = 0, rpt = 3, status_words = 0;
    bool seg_mode = true;            // wave-autonomous segmented compaction (cfg.compaction == 0)
    bool cull = true;                // AABB candidate culling in front of the exact tests (cfg.culling == 0)
    bool queue = false;              // typed work-queue kernel (cfg.ordering == 1; LDS geometry, G <= 32, no merging)
    bool pathq = false;              //   cfg.ordering == 2: whole paths in one launch (k_path_q), rings instead of pools
    float *d_arena = nullptr;        //   [grid_path * kWaves][kSFields][kStack]: the waves' ray stacks
    uint32_t *d_tickets = nullptr;   //   [kTicketCtrs][kTicketStride]
    size_t arena_bytes = 0;
    int grid_path = 0;
    uint32_t path_static_eighths = 4;
    uint32_t lds_path = 0;
    bool flat_pool = false;          //   PT_FLAT_POOL=1 (read at upload): 64-bit flat addressing of the pool even below 4 GiB (A/B switch)
    bool queue_mesh = false;         //   its variant with mesh traversal (scene has MESH primitives with triangles)
    FaceFrame *d_frames = nullptr;   // [G][3] shading frames of the box primitives (k_bounce_q)
    CullRec *d_cull = nullptr;       // bounds for its culling pass, cubes first
    int q_nbox = 0, q_nsph = 0;
    // MESH primitives (pt_set_meshes): host copies, and one device blob [nodes | triangles] per mesh of the uploaded scene
    struct HostMesh { int geom_index; std::vector<float> v; std::vector<int> idx; };
    std::vector<HostMesh> meshes;
    std::vector<void *> d_mesh_blobs;
    uint32_t nseg = 0, seg_slots = 0;     // level 0 (what k_generate fills)
    uint32_t lvl_slots[66] = {0}, lvl_nseg[66] = {0};   // level entering bounce b
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
    if (c->d_status) (void)hipFree(c->d_status);
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
    if (c->d_tickets) (void)hipFree(c->d_tickets);
    c->d_tickets = nullptr;
    for (void *b : c->d_mesh_blobs) (void)hipFree(b);
    c->d_mesh_blobs.clear();
    if (c->image == c->image_own) c->image = nullptr;
    c->image_own = nullptr; c->d_geoms = nullptr; c->d_mats = nullptr; c->d_status = nullptr; c->d_display = nullptr;
    c->scene_ready = false;
}

// Every clear of device memory goes through hipMemsetAsync on the context's stream: that stream is
// created non-blocking, so a hipMemset on the null stream would NOT be ordered against the kernels
// launched here (it once wiped the primary-hit hook's output after the kernel had written it).
// Conservative world-space AABB of a primitive for the culling pass (double precision, then
// inflated).  Box: the 8 transformed corners of [-.5,.5]^3.  Sphere (an ellipsoid after the affine
// map): centre +- 0.5*|row_k of the linear part|.  The inflation has to stay below RAY_BIAS_AMOUNT
// (2e-4) at scene scale, or every scattered ray would keep the wall it just left as a candidate;
// it has to exceed the few-ulp slop of the exact object-space tests (about 1e-6 at |x| ~ 10).
void world_bounds(const pt_geom &src, GeomRec *dst) {
    double lo[3], hi[3], maxabs = 0.0, maxrow = 0.0;
    const float *m = src.transform;
    for (int k = 0; k < 3; ++k) {
        const double a = m[4 * k], b = m[4 * k + 1], c3 = m[4 * k + 2], t = m[4 * k + 3];
        double ext;
        if (src.type == 0) ext = 0.5 * std::sqrt(a * a + b * b + c3 * c3);
        else ext = 0.5 * (std::fabs(a) + std::fabs(b) + std::fabs(c3));
        lo[k] = t - ext; hi[k] = t + ext;
        maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        maxrow = std::fmax(maxrow, std::sqrt(a * a + b * b + c3 * c3));
    }
    const double infl = 3e-5 + 4e-6 * maxabs;
    if (src.type == 0) {
        // bounding sphere of the ellipsoid: centre, R = largest semi-axis <= 0.5 * largest row norm...
        // the exact bound is 0.5 * the largest singular value; 0.5 * Frobenius norm is a safe over-estimate
        // for non-uniform scales and equals 0.5*s*sqrt(3) only for... keep it tight for the common
        // uniform case: rows of equal norm and orthogonal -> R = 0.5 * row norm.
        double fro = 0.0, ortho = 0.0;
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) fro += (double)m[4 * k + j] * m[4 * k + j];
        for (int k = 0; k < 3; ++k)
            for (int l = k + 1; l < 3; ++l) {
                double dotp = 0.0;
                for (int j = 0; j < 3; ++j) dotp += (double)m[4 * k + j] * m[4 * l + j];
                ortho = std::fmax(ortho, std::fabs(dotp));
            }
        double r0 = 0.0, r1 = 1e300;
        for (int k = 0; k < 3; ++k) {
            const double rn = std::sqrt((double)m[4 * k] * m[4 * k] + (double)m[4 * k + 1] * m[4 * k + 1] + (double)m[4 * k + 2] * m[4 * k + 2]);
            r0 = std::fmax(r0, rn); r1 = std::fmin(r1, rn);
        }
        const bool uniform = (r0 - r1) <= 1e-5 * r0 && ortho <= 1e-5 * r0 * r0;
        const double R = (uniform ? 0.5 * r0 : 0.5 * std::sqrt(fro)) + infl;
        dst->bmin[0] = m[3]; dst->bmin[1] = m[7]; dst->bmin[2] = m[11];
        dst->bmin[3] = std::nextafterf((float)(R * R * (1.0 + 1e-5)), INFINITY);
        dst->bmax[0] = dst->bmax[1] = dst->bmax[2] = 0.0f;
        dst->bmax[3] = std::nextafterf((float)R, INFINITY);
    } else {
        for (int k = 0; k < 3; ++k) {
            dst->bmin[k] = std::nextafterf((float)(lo[k] - infl), -INFINITY);
            dst->bmax[k] = std::nextafterf((float)(hi[k] + infl), INFINITY);
        }
        dst->bmin[3] = dst->bmax[3] = 0.0f;
    }
    // the sphere test reports the point 1e-4 (object space, along the ray) in front of the surface
    dst->slack = src.type == 0 ? (float)(1.5e-4 * maxrow + 1e-5) : 1e-5f;
}

// ---- MESH: threaded BVH over the triangles of one mesh (object space), built at upload ----------------------
// Median split of the triangle centroids along the widest axis, <= 4 triangles per leaf, nodes in depth-first
// order with skip links (traversal needs no stack).  Boxes are the exact float min/max of the member vertices,
// inflated by 1e-5 * (1 + largest |coordinate|): the slab test adds its own relative margins (cull_box).
struct MeshBuild {
    const float *v;
    const int *idx;
    std::vector<int> order;                  // triangle permutation (leaf ranges index into it)
    std::vector<MeshNode> nodes;
    std::vector<float> cen;                  // 3 per triangle
    void bounds(int first, int count, float lo[3], float hi[3]) const {
        for (int k = 0; k < 3; ++k) { lo[k] = 3e38f; hi[k] = -3e38f; }
        for (int i = first; i < first + count; ++i)
            for (int c = 0; c < 3; ++c) {
                const float *p = v + 3 * idx[3 * order[i] + c];
                for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], p[k]); hi[k] = std::fmax(hi[k], p[k]); }
            }
    }
    int emit(int first, int count, int parent_skip) {
        const int id = (int)nodes.size();
        nodes.emplace_back();
        float lo[3], hi[3];
        bounds(first, count, lo, hi);
        float maxabs = 0.0f;
        for (int k = 0; k < 3; ++k) maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        const float infl = 1e-5f * (1.0f + maxabs);
        for (int k = 0; k < 3; ++k) { nodes[id].bmin[k] = lo[k] - infl; nodes[id].bmax[k] = hi[k] + infl; }
        nodes[id].skip = parent_skip;
        if (count <= 4) { nodes[id].leaf = first | (count << 27); return id; }
        nodes[id].leaf = -1;
        int axis = 0;
        float ext = -1.0f;
        for (int k = 0; k < 3; ++k) {
            float cmin = 3e38f, cmax = -3e38f;
            for (int i = first; i < first + count; ++i) { cmin = std::fmin(cmin, cen[3 * order[i] + k]); cmax = std::fmax(cmax, cen[3 * order[i] + k]); }
            if (cmax - cmin > ext) { ext = cmax - cmin; axis = k; }
        }
        const int half = count / 2;
        std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count,
                         [&](int x, int y) { return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y); });
        const int left = emit(first, half, -2);                      // -2: "the right sibling", known once the left subtree is out
        const int right = emit(first + half, count - half, parent_skip);
        for (int k = left; k < right; ++k)
            if (nodes[k].skip == -2) nodes[k].skip = right;
        return id;
    }
};

// [MeshNode x nnodes (padded to a multiple of 2) | MeshTri x ntris] for one mesh; *tri_offset = byte offset of the triangles
std::vector<unsigned char> build_mesh_blob(const pt_context::HostMesh &hm, uint32_t *tri_offset) {
    MeshBuild mb;
    mb.v = hm.v.data(); mb.idx = hm.idx.data();
    const int nt = (int)(hm.idx.size() / 3);
    mb.order.resize(nt); mb.cen.resize((size_t)3 * nt);
    for (int t = 0; t < nt; ++t) {
        mb.order[t] = t;
        for (int k = 0; k < 3; ++k)
            mb.cen[3 * t + k] = (hm.v[3 * hm.idx[3 * t] + k] + hm.v[3 * hm.idx[3 * t + 1] + k] + hm.v[3 * hm.idx[3 * t + 2] + k]) * (1.0f / 3.0f);
    }
    mb.emit(0, nt, -1);
    const size_t nn = (mb.nodes.size() + 1) & ~(size_t)1;
    *tri_offset = (uint32_t)(nn * sizeof(MeshNode));
    std::vector<unsigned char> blob(nn * sizeof(MeshNode) + (size_t)nt * sizeof(MeshTri), 0);
    memcpy(blob.data(), mb.nodes.data(), mb.nodes.size() * sizeof(MeshNode));
    MeshTri *tris = reinterpret_cast<MeshTri *>(blob.data() + *tri_offset);
    for (int i = 0; i < nt; ++i) {
        const int t = mb.order[i];
        const float *p0 = &hm.v[3 * hm.idx[3 * t]], *p1 = &hm.v[3 * hm.idx[3 * t + 1]], *p2 = &hm.v[3 * hm.idx[3 * t + 2]];
        const f3 v0 = mk(p0[0], p0[1], p0[2]);
        const f3 e1 = mk(p1[0], p1[1], p1[2]) - v0, e2 = mk(p2[0], p2[1], p2[2]) - v0;     // the kernels' own float subtraction
        const f3 ng = cross(e1, e2);
        MeshTri &r = tris[i];
        r.v0[0] = v0.x; r.v0[1] = v0.y; r.v0[2] = v0.z; r.index = t;
        r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z;
        r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
        r.ng[0] = ng.x; r.ng[1] = ng.y; r.ng[2] = ng.z;
    }
    return blob;
}

// conservative world-space AABB of a mesh primitive (double precision, inflated like the cubes')
void mesh_world_bounds(const pt_geom &src, const pt_context::HostMesh &hm, GeomRec *dst) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, maxabs = 0.0, maxrow = 0.0;
    const float *m = src.transform;
    for (size_t i = 0; i + 2 < hm.v.size(); i += 3)
        for (int k = 0; k < 3; ++k) {
            const double w = (double)m[4 * k] * hm.v[i] + (double)m[4 * k + 1] * hm.v[i + 1] + (double)m[4 * k + 2] * hm.v[i + 2] + (double)m[4 * k + 3];
            lo[k] = std::fmin(lo[k], w); hi[k] = std::fmax(hi[k], w);
        }
    for (int k = 0; k < 3; ++k) {
        maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        maxrow = std::fmax(maxrow, std::sqrt((double)m[4 * k] * m[4 * k] + (double)m[4 * k + 1] * m[4 * k + 1] + (double)m[4 * k + 2] * m[4 * k + 2]));
    }
    const double infl = 3e-5 + 4e-6 * maxabs;
    for (int k = 0; k < 3; ++k) {
        dst->bmin[k] = std::nextafterf((float)(lo[k] - infl), -INFINITY);
        dst->bmax[k] = std::nextafterf((float)(hi[k] + infl), INFINITY);
    }
    dst->slack = (float)(1.5e-4 * maxrow + 1e-5);        // the hit point sits 1e-4 (object space) in front of the surface
}

template <bool LDS, bool LAST>
int launch_bounce_t(pt_context *c, const BounceArgs &a) {
    hipLaunchKernelGGL((k_bounce<LDS, LAST>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                       (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

int launch_bounce(pt_context *c, const BounceArgs &a, bool last) {
    Scoped s(c, 1);
    if (c->geom_lds) return last ? launch_bounce_t<true, true>(c, a) : launch_bounce_t<true, false>(c, a);
    return last ? launch_bounce_t<false, true>(c, a) : launch_bounce_t<false, false>(c, a);
}

template <bool LDS, bool LAST, bool CULL, bool GEN>
int launch_seg_t(pt_context *c, const SegArgs &a) {
    hipLaunchKernelGGL((k_bounce_seg<LDS, LAST, CULL, GEN>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                       (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

template <bool LDS, bool CULL>
int launch_seg_lc(pt_context *c, const SegArgs &a, bool last, bool gen) {
    if (gen) return last ? launch_seg_t<LDS, true, CULL, true>(c, a) : launch_seg_t<LDS, false, CULL, true>(c, a);
    return last ? launch_seg_t<LDS, true, CULL, false>(c, a) : launch_seg_t<LDS, false, CULL, false>(c, a);
}

template <bool LAST, bool GEN>
int launch_wide_t(pt_context *c, const SegArgs &a) {
    hipLaunchKernelGGL((k_bounce_seg<true, LAST, true, GEN, false, true>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                       (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

template <bool LAST, bool GEN>
int launch_nee_t(pt_context *c, const SegArgs &a) {
    hipLaunchKernelGGL((k_bounce_seg<true, LAST, true, GEN, true>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                       (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

template <bool LAST, bool GEN>
int launch_q_t(pt_context *c, const SegArgs &a) {
    QTables qt;
    qt.frames = c->d_frames; qt.cull = c->d_cull; qt.nbox = c->q_nbox; qt.nsph = c->q_nsph;
    if (c->queue_mesh)
        hipLaunchKernelGGL((k_bounce_q<LAST, GEN, true>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                           (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats, qt);
    else
        hipLaunchKernelGGL((k_bounce_q<LAST, GEN>), dim3(c->grid_bounce), dim3(kBlock), c->lds_bytes, c->stream, a,
                           (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats, qt);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

int launch_seg(pt_context *c, const SegArgs &a, bool last, bool gen) {
    Scoped s(c, 1);
    if (c->queue) {
        if (gen) return last ? launch_q_t<true, true>(c, a) : launch_q_t<false, true>(c, a);
        return last ? launch_q_t<true, false>(c, a) : launch_q_t<false, false>(c, a);
    }
    if (c->nee) {
        if (gen) return last ? launch_nee_t<true, true>(c, a) : launch_nee_t<false, true>(c, a);
        return last ? launch_nee_t<true, false>(c, a) : launch_nee_t<false, false>(c, a);
    }
    if (c->wide) {
        if (gen) return last ? launch_wide_t<true, true>(c, a) : launch_wide_t<false, true>(c, a);
        return last ? launch_wide_t<true, false>(c, a) : launch_wide_t<false, false>(c, a);
    }
    if (c->cull) return c->geom_lds ? launch_seg_lc<true, true>(c, a, last, gen) : launch_seg_lc<false, true>(c, a, last, gen);
    return c->geom_lds ? launch_seg_lc<true, false>(c, a, last, gen) : launch_seg_lc<false, false>(c, a, last, gen);
}

// segment levels for a launch group of n_rays rays: level b = layout of the pool entering bounce b
// Segment size for a launch group: fixed by cfg.chunk_rays, else about four segments per resident
// wave, a multiple of 64 (full wave groups) between 192 and 1024 -- small launches need many small
// segments to occupy every wave, big (batched) launches run best on long ones (measured).
uint32_t seg_slots_for(const pt_context *c, uint32_t n_rays) {
    if (c->cfg.chunk_rays > 0) return c->seg_slots;
    // the sparse-work queue drains once per segment (one partly filled group): longer segments there
    const bool longseg = c->queue;
    const uint32_t slots = (uint32_t)c->grid_bounce * kWaves * (longseg ? 2u : 4u);
    uint32_t S = (((n_rays + slots - 1) / slots) + 63u) & ~63u;
    if (S < 192u) S = 192u;
    if (S > (longseg ? 2048u : 1024u)) S = longseg ? 2048u : 1024u;
    return S;
}

void plan_levels(const pt_context *c, uint32_t n_rays, uint32_t *slots, uint32_t *nseg) {
    const uint32_t S = seg_slots_for(c, n_rays);
    slots[0] = S; nseg[0] = (n_rays + S - 1) / S;
    const uint32_t floor_segs = (uint32_t)(c->cfg.merge_floor > 0 ? c->cfg.merge_floor : 0);
    for (int b = 0; b < c->cfg.max_depth; ++b) {
        const uint32_t half = (nseg[b] + 1u) / 2u;
        const bool merge = c->cfg.merge_floor > 0 && half >= floor_segs && slots[b] * 2u <= 65536u;
        slots[b + 1] = merge ? slots[b] * 2u : slots[b];
        nseg[b + 1] = merge ? half : nseg[b];
    }
}

// `batch` consecutive iterations starting at `iteration` as ONE launch group (batch > 1: segmented
// path only).  stop_after < 0 renders all bounces, otherwise only the first `stop_after` bounces
// without the LAST variant (parity hook, batch = 1).
int enqueue_iterations(pt_context *c, uint32_t iteration, uint32_t batch, int stop_after) {
    const int D = c->cfg.max_depth;
    // The segmented path generates camera rays inside its first bounce launch; k_generate runs only
    // for the look-back variant and for the parity hook that wants the pool before any bounce.
    const bool fused = c->seg_mode && stop_after != 0;
    const uint32_t n_rays = batch * c->n_own;
    if (c->seg_mode) plan_levels(c, n_rays, c->lvl_slots, c->lvl_nseg);
    if (!fused) {
        Scoped s(c, 0);
        GenArgs g;
        g.cam = c->cam; g.pool = c->pool[0]; g.cap = c->cap; g.n_own = c->n_own; g.iteration = iteration;
        g.sync = c->d_sync; g.status = c->d_status; g.status_words = c->seg_mode ? 0u : c->status_words; g.depth = D;
        g.seg_cnt0 = c->seg_mode ? c->d_segcnt[0] : nullptr; g.nseg = c->lvl_nseg[0]; g.seg_slots = c->lvl_slots[0];
        uint32_t work = c->n_own > g.status_words ? c->n_own : g.status_words;
        if (c->seg_mode && g.nseg > work) work = g.nseg;
        hipLaunchKernelGGL(k_generate, dim3((work + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, g);
        HIPCHK(hipGetLastError());
    }
    const int nb = stop_after < 0 ? D : stop_after;
    if (fused) c->bank ^= 1u;
    if (c->pathq && stop_after < 0) {
        // whole paths: one launch for the group (the per-bounce launches below remain the parity hooks' path)
        SegArgs a;
        memset(&a, 0, sizeof a);
        a.cap = c->cap; a.image = c->image; a.G = c->G; a.M = c->M; a.sync = c->d_sync;
        a.iteration = iteration; a.n_own = c->n_own; a.cam = c->cam; a.bank = c->bank; a.pix_mask = c->pix_mask;
        a.n_rays = n_rays; a.batch = batch; a.planes = c->d_planes; a.plane_stride = (size_t)c->n_own * 3;
        PathArgs pa;
        pa.arena = c->d_arena; pa.arena_bytes = c->arena_bytes < (1ull << 32) ? (uint32_t)c->arena_bytes : 0u;
        pa.depth = (uint32_t)D; pa.ticket = c->d_tickets; pa.error = &c->d_sync->error;
        {   // job size: about 48 jobs per wave of the launch, whole groups, 64 .. kJobMax rays
            const uint64_t per_wave = (uint64_t)n_rays / ((uint64_t)c->grid_path * kWaves * 48u);
            uint32_t job = (uint32_t)((per_wave + 63u) & ~63ull);
            if (job < 64u) job = 64u;
            if (job > kJobMax) job = kJobMax;
            if (c->cfg.chunk_rays > 0) job = (uint32_t)((c->cfg.chunk_rays + 63) & ~63);      // explicit
            pa.job_rays = job;
            // static share: half of the jobs (PT_P_STATIC_EIGHTHS = 0..8, read at upload: A/B switch and test hook)
            const uint64_t njobs = ((uint64_t)n_rays + job - 1) / job, waves = (uint64_t)c->grid_path * kWaves;
            pa.static_rounds = (uint32_t)(njobs * c->path_static_eighths / 8u / waves);
        }
        QTables qt;
        qt.frames = c->d_frames; qt.cull = c->d_cull; qt.nbox = c->q_nbox; qt.nsph = c->q_nsph;
        HIPCHK(hipMemsetAsync(pa.ticket, 0, (size_t)kTicketCtrs * kTicketStride * sizeof(uint32_t), c->stream));
        {
            Scoped s(c, 1);
            if (c->queue_mesh)
                hipLaunchKernelGGL(k_path_q<true>, dim3(c->grid_path), dim3(kBlock), c->lds_path, c->stream, a, pa,
                                   (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats, qt);
            else
                hipLaunchKernelGGL(k_path_q<false>, dim3(c->grid_path), dim3(kBlock), c->lds_path, c->stream, a, pa,
                                   (const GeomRec *)c->d_geoms, (const MatRec *)c->d_mats, qt);
            HIPCHK(hipGetLastError());
        }
        if (batch > 1u) {
            Scoped s(c, 0);
            FoldArgs f;
            f.image = c->image; f.planes = c->d_planes; f.plane_stride = (size_t)c->n_own * 3;
            f.batch = batch; f.n_own = c->n_own; f.W = c->W; f.row_offset = c->cfg.row_offset; f.row_stride = c->cfg.row_stride;
            hipLaunchKernelGGL(k_fold, dim3((c->n_own + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, f);
            HIPCHK(hipGetLastError());
        }
        c->counts_pending = true;
        return PT_OK;
    }
    for (int b = 0; b < nb && c->seg_mode; ++b) {
        SegArgs a;
        a.in = c->pool[b & 1]; a.out = c->pool[(b + 1) & 1]; a.cap = c->cap; a.image = c->image;
        a.G = c->G; a.M = c->M; a.sync = c->d_sync;
        a.cnt_in = c->d_segcnt[b & 1]; a.cnt_out = c->d_segcnt[(b + 1) & 1];
        a.nseg_in = c->lvl_nseg[b]; a.nseg_out = c->lvl_nseg[b + 1]; a.seg_slots = c->lvl_slots[b];
        a.merge = c->lvl_slots[b + 1] != c->lvl_slots[b] ? 1u : 0u;
        a.bounce = b; a.iteration = iteration; a.n_own = c->n_own; a.cam = c->cam; a.bank = c->bank;
        a.pix_mask = c->pix_mask;
        { const uint64_t pb = (uint64_t)c->cap * kFields * sizeof(float); a.pool_bytes = (c->queue && pb < (1ull << 32) && !c->flat_pool) ? (uint32_t)pb : 0u; }
        a.n_rays = n_rays; a.batch = batch; a.planes = c->d_planes; a.plane_stride = (size_t)c->n_own * 3;
        a.lights = c->d_lights; a.nlights = c->nlights;
        a.nbc = c->nbc; a.nsc = c->nsc; a.cluster_bytes = c->wide ? c->cluster_bytes : 0u;
        const bool last = (stop_after < 0) && (b == D - 1);
        int rc = launch_seg(c, a, last, b == 0);
        if (rc) return rc;
    }
    if (c->seg_mode && (batch > 1u || c->nee)) {
        Scoped s(c, 0);
        FoldArgs f;
        f.image = c->image; f.planes = c->d_planes; f.plane_stride = (size_t)c->n_own * 3;
        f.batch = batch; f.n_own = c->n_own; f.W = c->W; f.row_offset = c->cfg.row_offset; f.row_stride = c->cfg.row_stride;
        hipLaunchKernelGGL(k_fold, dim3((c->n_own + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, f);
        HIPCHK(hipGetLastError());
    }
    for (int b = 0; b < nb && !c->seg_mode; ++b) {
        BounceArgs a;
        a.in = c->pool[b & 1]; a.out = c->pool[(b + 1) & 1]; a.cap = c->cap; a.image = c->image;
        a.G = c->G; a.M = c->M;
        a.sync = c->d_sync; a.status = c->d_status + (size_t)b * c->max_chunks;
        a.rpt = c->rpt; a.bounce = b; a.iteration = iteration;
        const bool last = (stop_after < 0) && (b == D - 1);
        int rc = launch_bounce(c, a, last);
        if (rc) return rc;
    }
    c->counts_pending = true;
    return PT_OK;
}

int check_device_error(pt_context *c) {
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, &c->d_sync->error, sizeof err, hipMemcpyDeviceToHost));
    if (err == 2u || err == 3u) { pth::set_error("whole-path kernel: %s (device state corrupt)", err == 2u ? "a level ring overflowed" : "turn limit reached"); return PT_ERR_HIP; }
    if (err) { pth::set_error("compaction look-back exceeded its spin limit (device sync state corrupt)"); return PT_ERR_HIP; }
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
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        pth::set_error("pt_create: no HIP device visible (this library has no CPU fallback)");
        return PT_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= n) { pth::set_error("pt_create: device %d out of range (%d visible)", cfg->device, n); return PT_ERR_ARGUMENT; }
    if (cfg->max_depth < 1 || cfg->max_depth > 64) { pth::set_error("pt_create: max_depth %d not in 1..64", cfg->max_depth); return PT_ERR_ARGUMENT; }
    if (cfg->row_stride < 1 || cfg->row_offset < 0 || cfg->row_offset >= cfg->row_stride) { pth::set_error("pt_create: bad row_offset/row_stride"); return PT_ERR_ARGUMENT; }
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pth::set_error("pt_create: device %d is %s; this library carries gfx950 code objects only", cfg->device, prop.gcnArchName);
        return PT_ERR_NO_DEVICE;
    }
    if (cfg->streams > 1 && cfg->mode == 0) {
        if (cfg->streams > 8) { pth::set_error("pt_create: streams %d not in 1..8", cfg->streams); return PT_ERR_ARGUMENT; }
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
    c->geom_lds = (cfg->geometry_path == 0);
    int chunk = cfg->chunk_rays > 0 ? cfg->chunk_rays : 768;
    c->rpt = (uint32_t)((chunk + kBlock - 1) / kBlock);
    if (c->rpt < 1) c->rpt = 1;
    if (c->rpt > 8) c->rpt = 8;
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
    std::vector<pt_context::HostMesh> copy;
    for (int i = 0; i < nmeshes; ++i) {
        const pt_mesh &m = meshes[i];
        if (!m.vertices || !m.indices || m.nvertices < 3 || m.ntriangles < 1 || m.ntriangles >= (1 << 27) || m.geom_index < 0) {
            pth::set_error("pt_set_meshes: mesh %d is empty or malformed", i);
            return PT_ERR_ARGUMENT;
        }
        for (int k = 0; k < 3 * m.ntriangles; ++k)
            if (m.indices[k] < 0 || m.indices[k] >= m.nvertices) { pth::set_error("pt_set_meshes: mesh %d: vertex index %d out of range (%d vertices)", i, m.indices[k], m.nvertices); return PT_ERR_ARGUMENT; }
        pt_context::HostMesh hm;
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
        world_bounds(geoms[i], &g[i]);
    }
    // MESH primitives with registered data (the others are skipped like the reference's empty branch)
    std::vector<const pt_context::HostMesh *> mesh_of(G, nullptr);
    bool have_mesh = false;
    for (const pt_context::HostMesh &hm : c->meshes) {
        if (hm.geom_index >= G || geoms[hm.geom_index].type != 2) { pth::set_error("pt_upload_scene: mesh registered for geom %d, which is not a MESH of this scene", hm.geom_index); return PT_ERR_ARGUMENT; }
        mesh_of[hm.geom_index] = &hm;
        have_mesh = true;
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
    c->seg_mode = (c->cfg.compaction == 0);
    c->cull = (c->cfg.culling == 0) && c->seg_mode;
    c->queue = c->cull && (c->cfg.ordering == 1 || c->cfg.ordering == 2) && c->cfg.geometry_path == 0 && G <= 32 && c->cfg.mode == 0 && c->cfg.merge_floor <= 0;
    c->queue_mesh = false;
    c->flat_pool = getenv("PT_FLAT_POOL") != nullptr;
    c->geom_lds = (c->cfg.geometry_path == 0);
    if (have_mesh) {
        // meshes: stable kernels and the typed work queues (where they share the spheres' stack)
        c->queue_mesh = c->queue;
        for (int i = 0; i < G; ++i) {
            if (!mesh_of[i]) continue;
            uint32_t tri_offset = 0;
            const std::vector<unsigned char> blob = build_mesh_blob(*mesh_of[i], &tri_offset);
            void *d_blob = nullptr;
            HIPCHK(hipMalloc(&d_blob, blob.size()));
            c->d_mesh_blobs.push_back(d_blob);
            HIPCHK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
            const unsigned long long addr = (unsigned long long)(uintptr_t)d_blob;
            const uint32_t lo32 = (uint32_t)addr, hi32 = (uint32_t)(addr >> 32);
            mesh_world_bounds(geoms[i], *mesh_of[i], &g[i]);
            memcpy(&g[i].bmin[3], &lo32, 4);
            memcpy(&g[i].bmax[3], &hi32, 4);
            g[i].inside_hits = (int)tri_offset;
        }
    }
    c->nee = c->cfg.direct_light != 0 && c->cfg.mode == 0;
    if (c->nee) {
        // one kernel family implements it: segmented compaction, culling, LDS tables, generation order
        if (!c->seg_mode || !c->cull || !c->geom_lds) {
            pth::set_error("pt_upload_scene: direct_light needs compaction=0, culling=0, geometry_path=0");
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

    // LDS budget: tables (+ the ray stage of the look-back variant)
    uint32_t tb = tables_bytes(G, M, c->geom_lds);
    const uint32_t stage_bytes = c->seg_mode ? (c->queue ? q_lds_offset(G, M) - tables_bytes(G, M, true) + kWaves * kQCap * kQFields * (uint32_t)sizeof(float) : 0u)
                                             : kBlock * c->rpt * kFields * (uint32_t)sizeof(float);
    if (c->geom_lds && tb + stage_bytes > 160u * 1024u) {   // table too large for LDS: scalar-load path
        if (c->nee) { pth::set_error("pt_upload_scene: direct_light needs the geometry table in LDS (%d primitives do not fit)", G); return PT_ERR_ARGUMENT; }
        c->geom_lds = false;
        tb = tables_bytes(G, M, false);
    }
    c->lds_bytes = tb + stage_bytes;
    // 33..256 primitives with the table in LDS: the mask-register / packed-list variant (PT_WIDE=0 turns it off)
    c->wide = c->cull && c->geom_lds && !c->nee && !c->queue && !have_mesh && c->cfg.mode == 0 && G > 32 && G <= 256;
    if (const char *wv = getenv("PT_WIDE")) if (atoi(wv) == 0) c->wide = false;
    // two-level culling of the many-primitive variant: clusters of <= kClusterSize primitives of one type
    std::vector<unsigned char> cluster_blob;
    c->nbc = c->nsc = 0; c->cluster_bytes = 0;
    if (c->wide) {
        std::vector<ClusterRec> recs;
        std::vector<unsigned char> ids;
        int csize = PT_CLUSTER;
        if (const char *cv = getenv("PT_CLUSTER_SIZE")) csize = atoi(cv);
        if (csize < 1) csize = 1;
        if (csize > kClusterMax) csize = kClusterMax;
        for (; csize <= kClusterMax; ++csize) {                // the per-lane cluster mask has 64 bits
            int nb = 0, ns = 0;
            for (int i = 0; i < G; ++i) { if (g[i].type == 1) nb++; else if (g[i].type == 0) ns++; }
            if ((nb + csize - 1) / csize + (ns + csize - 1) / csize <= 64) break;
        }
        if (csize > kClusterMax) csize = kClusterMax;
        for (int pass = 0; pass < 2; ++pass) {
            const int type = pass == 0 ? 1 : 0;
            std::vector<int> prim;
            for (int i = 0; i < G; ++i) if (g[i].type == type) prim.push_back(i);
            auto lo_of = [&](int i, int k) { return type == 1 ? g[i].bmin[k] : g[i].bmin[k] - g[i].bmax[3]; };
            auto hi_of = [&](int i, int k) { return type == 1 ? g[i].bmax[k] : g[i].bmin[k] + g[i].bmax[3]; };
            // recursive median split of the centres along the widest axis, left parts whole numbers of clusters
            std::vector<std::pair<int, int>> stack{{0, (int)prim.size()}};
            std::vector<std::pair<int, int>> leaves;
            while (!stack.empty()) {
                const std::pair<int, int> r = stack.back();
                stack.pop_back();
                const int first = r.first, count = r.second;
                if (count <= 0) continue;
                if (count <= csize) { leaves.push_back(r); continue; }
                int axis = 0;
                float ext = -1.0f;
                for (int k = 0; k < 3; ++k) {
                    float cmin = 3e38f, cmax = -3e38f;
                    for (int q = first; q < first + count; ++q) { const float cc = lo_of(prim[q], k) + hi_of(prim[q], k); cmin = std::fmin(cmin, cc); cmax = std::fmax(cmax, cc); }
                    if (cmax - cmin > ext) { ext = cmax - cmin; axis = k; }
                }
                int half = ((count / 2 + csize - 1) / csize) * csize;
                if (half >= count) half = count - csize > 0 ? count - csize : count / 2;
                std::nth_element(prim.begin() + first, prim.begin() + first + half, prim.begin() + first + count, [&](int x, int y) {
                    const float cx = lo_of(x, axis) + hi_of(x, axis), cy = lo_of(y, axis) + hi_of(y, axis);
                    return cx < cy || (cx == cy && x < y);
                });
                stack.push_back({first + half, count - half});
                stack.push_back({first, half});
            }
            for (const std::pair<int, int> &lf : leaves) {
                ClusterRec r;
                for (int k = 0; k < 3; ++k) { r.bmin[k] = 3e38f; r.bmax[k] = -3e38f; }
                r.first = (int)ids.size(); r.count = lf.second;
                for (int q = lf.first; q < lf.first + lf.second; ++q) {
                    ids.push_back((unsigned char)prim[q]);
                    for (int k = 0; k < 3; ++k) { r.bmin[k] = std::fmin(r.bmin[k], lo_of(prim[q], k)); r.bmax[k] = std::fmax(r.bmax[k], hi_of(prim[q], k)); }
                }
                recs.push_back(r);
                if (pass == 0) c->nbc++; else c->nsc++;
            }
        }
        if (c->nbc + c->nsc > 64) c->wide = false;            // the per-lane cluster mask has 64 bits
        else {
            const size_t idbytes = (ids.size() + 15) & ~(size_t)15;
            cluster_blob.assign(recs.size() * sizeof(ClusterRec) + idbytes, 0);
            memcpy(cluster_blob.data(), recs.data(), recs.size() * sizeof(ClusterRec));
            memcpy(cluster_blob.data() + recs.size() * sizeof(ClusterRec), ids.data(), ids.size());
            c->cluster_bytes = (uint32_t)cluster_blob.size();
            c->lds_bytes += c->cluster_bytes;
        }
    }
    const void *fns[8] = {
        reinterpret_cast<const void *>(&k_bounce<true, false>), reinterpret_cast<const void *>(&k_bounce<false, false>),
        reinterpret_cast<const void *>(&k_bounce<true, true>), reinterpret_cast<const void *>(&k_bounce<false, true>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<true, false, true, false> : &k_bounce_seg<true, false, false, false>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<false, false, true, false> : &k_bounce_seg<false, false, false, false>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<true, true, true, false> : &k_bounce_seg<true, true, false, false>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<false, true, true, false> : &k_bounce_seg<false, true, false, false>)};
    const void *gen_fns[4] = {
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<true, false, true, true> : &k_bounce_seg<true, false, false, true>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<false, false, true, true> : &k_bounce_seg<false, false, false, true>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<true, true, true, true> : &k_bounce_seg<true, true, false, true>),
        reinterpret_cast<const void *>(c->cull ? &k_bounce_seg<false, true, true, true> : &k_bounce_seg<false, true, false, true>)};
    const void *nee_fns[4] = {
        reinterpret_cast<const void *>(&k_bounce_seg<true, false, true, false, true>), reinterpret_cast<const void *>(&k_bounce_seg<true, true, true, false, true>),
        reinterpret_cast<const void *>(&k_bounce_seg<true, false, true, true, true>), reinterpret_cast<const void *>(&k_bounce_seg<true, true, true, true, true>)};
    if (c->lds_bytes > 64u * 1024u) {
        const void *wide_fns[4] = {
            reinterpret_cast<const void *>(&k_bounce_seg<true, false, true, false, false, true>), reinterpret_cast<const void *>(&k_bounce_seg<true, true, true, false, false, true>),
            reinterpret_cast<const void *>(&k_bounce_seg<true, false, true, true, false, true>), reinterpret_cast<const void *>(&k_bounce_seg<true, true, true, true, false, true>)};
        for (const void *fn : wide_fns) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (const void *fn : nee_fns) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (const void *fn : fns) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (const void *fn : gen_fns) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }

    // persistent grid: CUs x resident blocks per CU
    int per_cu = c->cfg.blocks_per_cu;
    if (per_cu <= 0) {
        int occ = 0;
        const void *fn = c->wide ? reinterpret_cast<const void *>(&k_bounce_seg<true, false, true, false, false, true>)
                       : c->nee ? nee_fns[0]
                       : c->queue ? (c->queue_mesh ? reinterpret_cast<const void *>(&k_bounce_q<false, false, true>) : reinterpret_cast<const void *>(&k_bounce_q<false, false>))
                       : fns[(c->seg_mode ? 4 : 0) + (c->geom_lds ? 0 : 1)];
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kBlock, c->lds_bytes) != hipSuccess || occ < 1) occ = 2;
        per_cu = occ;
    }
    int grid = c->n_cu * per_cu;

    if (c->seg_mode) {
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
        uint32_t K = 1;
        if (c->cfg.mode == 0 && !big_frame) {
            if (c->cfg.batch > 0) K = (uint32_t)c->cfg.batch;
            else K = (uint32_t)((32u * 1024u * 1024u) / n_own);      // 16 at 1080p: measured best (14: +4 %, 18: +8 % time)
            // the slot field has 7 bits beside the count-emission flag; a plane holds the owned pixels: <= 4 GiB in all
            const uint64_t plane_bytes = (uint64_t)n_own * 3 * sizeof(float);
            const uint32_t by_memory = (uint32_t)((4ull << 30) / plane_bytes);
            if (K > 128u) K = 128u;
            if (K > by_memory) K = by_memory;
            if (K < 1u) K = 1u;
        }
        c->batch_max = K;
        const uint32_t max_rays = K * n_own;
        c->nseg = (max_rays + S - 1) / S;                 // S here = the smallest segment size in use
        c->cap = max_rays + 2u * (c->cfg.merge_floor > 0 ? 65536u : 4096u);
        c->max_chunks = c->nseg;
        if (K > 1u || c->nee) {
            HIPCHK(hipMalloc(&c->d_planes, (size_t)K * n_own * 3 * sizeof(float)));
            HIPCHK(hipMemsetAsync(c->d_planes, 0, (size_t)K * n_own * 3 * sizeof(float), c->stream));
        }
        const uint32_t blocks_needed = (c->nseg + kWaves - 1) / kWaves;
        if ((uint32_t)grid > blocks_needed) grid = (int)blocks_needed;
        if (grid < 1) grid = 1;
        c->grid_bounce = grid;
        plan_levels(c, max_rays, c->lvl_slots, c->lvl_nseg);
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc(&c->d_segcnt[i], (size_t)(c->nseg + 2u) * sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(c->d_segcnt[i], 0, (size_t)(c->nseg + 2u) * sizeof(uint32_t), c->stream));
        }
        c->status_words = 0;
    } else {
        const uint32_t chunk_rays = kBlock * c->rpt;
        c->max_chunks = (n_own + chunk_rays - 1) / chunk_rays;
        c->cap = c->max_chunks * chunk_rays;
        if ((uint32_t)grid > c->max_chunks) grid = (int)c->max_chunks;
        c->status_words = (uint32_t)c->cfg.max_depth * c->max_chunks;
        HIPCHK(hipMalloc(&c->d_status, (size_t)c->status_words * sizeof(u64)));
        HIPCHK(hipMemsetAsync(c->d_status, 0, (size_t)c->status_words * sizeof(u64), c->stream));
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
    if (c->queue) {
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
    c->pathq = c->queue && c->cfg.ordering == 2 && !c->nee;
    if (c->pathq) {
        c->lds_path = p_lds_bytes(G, M);
        int occ = 0;
        if (c->cfg.blocks_per_cu > 0) occ = c->cfg.blocks_per_cu;
        else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, c->queue_mesh ? reinterpret_cast<const void *>(&k_path_q<true>) : reinterpret_cast<const void *>(&k_path_q<false>), kBlock, c->lds_path) != hipSuccess || occ < 1) occ = 2;
        c->grid_path = c->n_cu * occ;
        c->arena_bytes = (size_t)c->grid_path * kWaves * (size_t)kSFields * kStack * sizeof(float);
        HIPCHK(hipMalloc(&c->d_arena, c->arena_bytes));
        HIPCHK(hipMalloc(&c->d_tickets, (size_t)kTicketCtrs * kTicketStride * sizeof(uint32_t)));
        c->path_static_eighths = 4u;
        if (const char *ev = getenv("PT_P_STATIC_EIGHTHS")) { const int v = atoi(ev); c->path_static_eighths = v < 0 ? 0u : v > 8 ? 8u : (uint32_t)v; }
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
            hipLaunchKernelGGL(k_flat, dim3((c->n_own + kBlock - 1) / kBlock), dim3(kBlock), tables_bytes(c->G, c->M, true), c->stream, f);
            HIPCHK(hipGetLastError());
            c->iterations++;
        } else {
            uint32_t b = (uint32_t)(first_iteration + count - it);
            if (b > c->batch_max) b = c->batch_max;
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
        hipLaunchKernelGGL(k_display, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->image, dst, n, scale);
        HIPCHK(hipGetLastError());
    }
    if (!out_is_device) {
        HIPCHK(hipMemcpyAsync(out, c->d_display, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return PT_OK;
}

#ifdef PT_CULL_STATS
int pt_debug_cull_stats(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cull_stats), 128) == hipSuccess ? 0 : -2;
}
#endif

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
    hipLaunchKernelGGL(k_flat, dim3((c->n_own + kBlock - 1) / kBlock), dim3(kBlock), tables_bytes(c->G, c->M, true), c->stream, f);
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
    c->image = scratch;
    int rc = enqueue_iterations(c, (uint32_t)iteration, 1u, bounces);
    c->image = saved;
    if (rc) { (void)hipFree(scratch); return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(scratch);
    SyncBlock after;
    HIPCHK(hipMemcpy(&after, c->d_sync, sizeof after, hipMemcpyDeviceToHost));
    const bool fused = c->seg_mode && bounces != 0;
    const uint32_t n = (fused && c->bank) ? after.counts_b[bounces] : after.counts[bounces];
    c->bank = bank_saved;
    HIPCHK(hipMemcpy(c->d_sync, &snapshot, sizeof snapshot, hipMemcpyHostToDevice));
    c->counts_pending = pending;
    if (after.error) { pth::set_error("compaction look-back exceeded its spin limit"); return PT_ERR_HIP; }
    if (count) *count = (int)n;
    const float *src = c->pool[bounces & 1];
    float *dst[9] = {ox, oy, oz, dx, dy, dz, tr, tg, tb};
    if (!c->seg_mode) {
        for (int f = 0; f < 9; ++f)
            if (dst[f] && n) HIPCHK(hipMemcpy(dst[f], src + (size_t)f * c->cap, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (pixel && n) HIPCHK(hipMemcpy(pixel, src + (size_t)9 * c->cap, (size_t)n * 4, hipMemcpyDeviceToHost));
    } else if (n) {
        // segments are dense prefixes in generation order: concatenate them
        const uint32_t nseg = c->lvl_nseg[bounces], slots = c->lvl_slots[bounces];
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

int pt_debug_rng_from_thread(pt_context *c, float resx, float resy, float time, int n, const int *xy, float *out3) {
    (void)resy;
    if (!c || n < 0 || !xy || !out3) { pth::set_error("pt_debug_rng_from_thread: bad argument"); return PT_ERR_ARGUMENT; }
    if (!c->subs.empty()) return pt_debug_rng_from_thread(c->subs[0], resx, resy, time, n, xy, out3);
    if (n == 0) return PT_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    int *d_xy = nullptr; float *d_out = nullptr;
    HIPCHK(hipMalloc(&d_xy, (size_t)n * 8)); HIPCHK(hipMalloc(&d_out, (size_t)n * 12));
    HIPCHK(hipMemcpy(d_xy, xy, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rng_from_thread, dim3((n + 255) / 256), dim3(256), 0, c->stream, resx, time, n, d_xy, d_out);
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
    hipLaunchKernelGGL(k_hemisphere, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_n, d_x, d_o);
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
    hipLaunchKernelGGL(k_light_points, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const GeomRec *)(c->d_geoms + geom), n, d_s, d_o);
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
    hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_a, d_s, d_c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(s, d_s, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(co, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_a); (void)hipFree(d_s); (void)hipFree(d_c);
    return PT_OK;
}

}  // extern "C"
