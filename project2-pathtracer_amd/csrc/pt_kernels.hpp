// pt_kernels.hpp -- what the kernel families (pt_k_*.hip) and the host side (pt_api.hip) share: launch argument
// blocks, the LDS table layout, the culling arithmetic and the nearest-hit loops, and the host-callable launchers
// each family exports.  One translation unit per family:
//
//   pt_k_seg.hip    k_bounce_seg   stable order (ordering = 0; the parity hooks' reference): one wave streams its pool segments 64 rays at a
//                                  time: [bounce 0: camera ray] -> candidate culling -> exact reference tests ->
//                                  scatter -> accumulate -> ballot/mbcnt compaction into the output segment.
//                                  Variants: NEE (direct light), WIDE (33..256 primitives), meshes.  + k_generate
//                                  (the pool before any bounce: parity hook only)
//   pt_k_queue.hip  k_bounce_q     ordering = 1, <= 32 primitives: two wave-private stages with LDS work queues by
//                                  candidate type, one launch per bounce
//   pt_k_path.hip   k_path_q       ordering = 2, <= 32 primitives: whole paths on the typed work queues, one launch
//                                  per group of iterations (pt_config_default: what bench.py, the adaptor and ptrender run)
//   pt_k_wide.hip   k_path_w       ordering = 2, more than 32 analytic primitives (any number): whole paths; a grid walk feeds dense (ray, cell) and
//                                  (ray, primitive) pairs, type-pure exact tests on full waves
//   pt_k_misc.hip   k_fold, k_flat (the reference kernel as shipped + primary-hit hook), k_display, KAT kernels
//   pt_api.hip      contexts, scene upload, launch planning, the C ABI of include/ptmi355.h
//   pt_build.cpp    host-only scene builders: culling bounds, k_path_w's grid, clusters, mesh BVHs (no device code)
//
// No CPU fallback lives anywhere here: every entry point needs a gfx950 device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptmi355.h"
#include "pt_device.hpp"

namespace ptk {

using namespace ptd;

constexpr int kBlock = 256;          // 4 waves
#ifndef PT_SEG_WAVES
#define PT_SEG_WAVES 6               // min waves per SIMD asked of the register allocator for the bounce kernels:
#endif                               // 80 VGPRs, <= 8 B of scratch; measured 4 % faster than 5 (83 VGPRs), 7 spills
constexpr int kWaves = kBlock / 64;
constexpr int kFields = 10;          // ox oy oz dx dy dz tr tg tb pixel

typedef unsigned long long u64;

struct SyncBlock {                   // device-resident, one per context
    uint32_t counts[72];             // live rays entering bounce k of the CURRENT iteration (bank 0)
    uint32_t counts_b[72];           // bank 1: the fused segmented path alternates banks per iteration
    u64 totals[72];                  // counts folded over finished iterations
    u64 emitted;                     // paths ended on an emitter
    uint32_t error;                  // set by a kernel that hit one of its guards (2: k_path_* stack overflow, 3: turn limit)
    uint32_t pad;
};

struct GenArgs {
    CamRec cam;
    float *pool;                     // field f at pool + f*cap
    uint32_t cap;
    uint32_t n_own;                  // rays this context generates per iteration
    uint32_t iteration;
    SyncBlock *sync;
    uint32_t *seg_cnt0;              // rays per segment entering bounce 0
    uint32_t nseg, seg_slots;
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

#ifdef PT_CULL_STATS
static __device__ unsigned long long g_cull_stats[16];  // [8..15] typed-queue kernel: fresh groups, fresh valid lanes, box groups, box lanes, sphere groups, sphere lanes, shaded lanes, re-queued lanes
__device__ __forceinline__ void qstat(int i, unsigned long long v) { if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g_cull_stats[i], v); }
// [0..7] lock-step kernels: [0] groups, [1] box iters, [2] box active lanes, [3] sph iters, [4] sph active lanes, [5] candidates
// whole-path kernel with meshes (tools/meshstats.py): [0] mesh_test calls (waves) [1] lanes in them [2] BVH node visits (wave trips)
// [3] lanes active in them [4] triangle tests (lanes)
#endif

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
    const int oct = (rd.x < 0.0f ? 1 : 0) | (rd.y < 0.0f ? 2 : 0) | (rd.z < 0.0f ? 4 : 0);      // near children first (MeshNode)
#ifdef PT_MESH_STATS
    { const unsigned long long act = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) { atomicAdd(&g_cull_stats[0], 1ull); atomicAdd(&g_cull_stats[1], (unsigned long long)__popcll(act)); } }
#endif
    while (node >= 0) {
#ifdef PT_MESH_STATS
        { const unsigned long long act = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) { atomicAdd(&g_cull_stats[2], 1ull); atomicAdd(&g_cull_stats[3], (unsigned long long)__popcll(act)); } }
#endif
        // (the blob is global memory: saying so gives global_load instead of flat_load, which also counts as an LDS access)
        typedef float nf4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) nf4 *gf4;
        typedef const __attribute__((address_space(1))) int *gi1;
        const gf4 np = (gf4)(uintptr_t)(nodes + node);
        const nf4 lo = np[0];                                                        // bmin.xyz, leaf
        const nf4 hi = np[1];                                                        // bmax.xyz, far
        const int skip = ((gi1)(uintptr_t)(nodes + node))[8 + oct];
        const float bl[3] = {lo.x, lo.y, lo.z}, bh[3] = {hi.x, hi.y, hi.z};
        float tn;
        const bool in = cull_box(bl, bh, cr, tn) && !(tn > best);
        const int leaf = __float_as_int(lo.w), far = __float_as_int(hi.w);
        if (!in) { node = skip; continue; }
        if (leaf < 0) { node = ((oct >> (far & 3)) & 1) ? (far >> 2) : node + 1; continue; }
        const int first = leaf & 0x7FFFFFF, cnt = (int)((uint32_t)leaf >> 27);
        for (int k = 0; k < cnt; ++k) {
#ifdef PT_MESH_STATS
            atomicAdd(&g_cull_stats[4], 1ull);
#endif
            const gf4 tp = (gf4)(uintptr_t)(tris + first + k);
            const nf4 a = tp[0], b = tp[1], c = tp[2];
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
constexpr int kClusterMax = 16;
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
    uint32_t nseg_in, nseg_out;      // segments of the launch group (the same layout on both sides)
    uint32_t seg_slots;              // S: slots per segment
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
// Records per wave: as many as leave the kernel's occupancy target standing (five blocks per CU) next to the
// scene's tables -- 144 on the Cornell box (138 measured 1.5 % slower, 146 costs the fifth block: 0.2078 vs 0.1656 ms/step).
// The kernel is instantiated for a few capacities; pt_upload_scene takes the largest that fits (path_pick_cap).
#define PT_PATH_CAPS(X) X(160) X(152) X(144) X(136) X(128) X(120) X(112) X(104) X(96) X(88) X(80)
constexpr int kPCaps[] = {160, 152, 144, 136, 128, 120, 112, 104, 96, 88, 80};
constexpr uint32_t kPCapMax = 160;
constexpr uint32_t kPFields = 12;        // ox oy oz dx dy dz tx ty tz pixelword mask candidate|level<<8
constexpr uint32_t kPParked = 0;         // (words of a record parked in the arena behind the wave's stack: the NEE variant adds its own)
#ifndef PT_STACK_SLOTS
#define PT_STACK_SLOTS 256               // a smaller value is a test build: it provokes the overflow guard (tests/test_gpu_round3.py)
#endif
constexpr uint32_t kStack = PT_STACK_SLOTS;   // rays on a wave's stack.  Bound = the wave's whole population: camera rays only enter while the stack holds
                                              //   fewer than 64 rays and the queues at most cap - 64 records: 63 + 96 + 64 = 223 rays in all at most, wherever they
                                              //   sit later (k_path_q<MESH>: + the kMStack rays of the mesh stack, so its stack is that much deeper)
constexpr uint32_t kSFields = 11;        // ox oy oz dx dy dz tx ty tz pixelword level
constexpr uint32_t kTicketCtrs = 16, kTicketStride = 64;
constexpr uint32_t kJobMax = 128;        // camera rays per job: about 1/48 of a wave's share of the launch, 64 .. kJobMax

// Scope of the fences between a wave's stores to its OWN arena (stack of survivors, parked words) and its later loads of them.
// "workgroup" makes the wave wait until the stores have landed (s_waitcnt vmcnt(0)) before it loads.
#ifndef PT_SELF_SCOPE
#define PT_SELF_SCOPE "workgroup"
#endif

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
    uint32_t turn_limit;             // scheduling turns a wave may take before it gives up with error 3 (the host prices it at the worst case of the launch x 4: pt_api.hip; tests lower it)
    float qscale, slack_max;         // k_path_w: candidate keys carry floor(entry distance * qscale); largest GeomRec::slack of the scene
    // parity hook (pt_debug_trace_pool with ordering = 2): rays that survive bounce tap_level - 1 are written here (10 fields, SoA,
    // stride tap_cap, in the order the waves meet them; the host sorts them by pixel) instead of going on to their next bounce
    float *tap;
    uint32_t *tap_count;
    uint32_t tap_level, tap_cap;     // tap_level 0: off (every render)
};
constexpr uint32_t kWPayload = 16;   // k_path_w: floats per payload record in global memory: one 64-byte line per ray (layout: pt_k_wide.hip)

// k_path_w's spatial index: a uniform grid over the SMALL analytic primitives (host: build_grid, pt_build.cpp).  A ray
// walks the cells it crosses (3D-DDA) and only the primitives listed there have their own bounds tested; primitives
// that would be listed in more than kGridBigCells cells (walls, floors) wait in a short list every ray tests.
// Blob, staged in LDS behind the geometry table: cells[ncells] (first ref | count << 16), refs[nrefs] (16 bits each),
// big[nbig] (primitive ids).  A ref is the primitive id | flags << 8: bits 0..2 this cell is the LOWEST cell of the
// primitive's cell range on x, y, z; bits 3..5 the HIGHEST; bit 6 it is the cell's last reference; bit 7 the primitive is a sphere.
// The flags make the walk list every primitive ONCE without any arithmetic on distances: the cells of a ray inside a
// primitive's (box-shaped) cell range are consecutive, so the primitive is new in a cell exactly when the step into
// that cell crossed the range's boundary on one of the axes stepped -- or when the cell is the ray's first.
constexpr int kGridBigCells = 27;
constexpr uint32_t kGridMaxCells = 8192;            // narrow references: 13 bits of a pair entry hold a reference index
// More than 256 primitives (k_path_w<BIG>): the same grid with WIDE references -- cells[ncells] hold the index of the cell's first
// reference (0xFFFFFFFF: empty), a reference is primitive id | flags << 24 (same flag bits), big[nbig] are 32-bit ids; the blob is
// staged in LDS when it fits beside the waves' regions and read from global memory otherwise.
constexpr uint32_t kGridMaxCellsWide = 1u << 18;    // (cell addresses are rebuilt in float: exact far beyond this)
constexpr uint32_t kGridMaxRefsWide = (1u << 18) - 1u;   // 18 bits of a pair entry hold a reference index
struct GridArgs {
    float gmin[3];                   // lower corner
    float h[3], inv_h[3];            // cell size per axis and its reciprocal
    int n[3];                        // cells per axis
    float centre[3], reach;          // a ray starting farther than `reach` from the centre on any axis is not walked (its
                                     //   float error would exceed the margin the cells were filled with): reference loop
    uint32_t ncells, nrefs, nbig;
    uint32_t blob_bytes;             // multiple of 16
    const unsigned char *blob;       // device copy
    uint32_t bin1, bin2;             // survivors wait for their next bounce sorted by the length of their walk in cells: <= bin1, <= bin2, longer
    uint32_t in_lds;                 // wide references only: the blob is staged in LDS (it fits beside the waves' regions); 0: read from `blob`
};
constexpr uint32_t kWalkBins = 3;

// One ray's walk through the grid: the state a lane keeps, how it starts and how it steps.  Host-callable: the grid
// probe of the CPU-side tests (pt_debug_grid_probe) runs these very functions over the host copy of the blob.
// `inv` is 1/d per axis with |d| clamped away from zero (make_cull_ray on the device); its SIGN says which way the
// axis steps.  The distance to the k-th boundary of an axis is fma(k, dt, t0) -- one rounding whatever k is, no
// accumulated error -- and the cell's address is rebuilt from the three step counters the same way (exact: small integers).
struct GridWalk {
    float tx, ty, tz;                // distance to the next cell boundary per axis
    float kx, ky, kz;                // steps taken per axis (integer-valued)
    float t0x, t0y, t0z, dtx, dty, dtz;   // tx = fma(kx, dtx, t0x)
    float kmx, kmy, kmz;             // steps the grid allows per axis: one more leaves it
    float c0, scx, scy, scz;         // linear cell index = c0 + kx * scx + ky * scy + kz * scz (signed strides)
    uint32_t selmask;                // the reference flags that mean "new" per axis: stepping up meets a cell range at its lowest
                                     //   cell (bits 0..2), stepping down at its highest (bits 3..5)
    uint32_t emask;                  // flags that make a reference new in the current cell; bit 6: the ray's first cell (all are new)
    bool walking;
};

// `wanted` rays only; the part of the ray inside the grid's box is [t0, t1] (margins on both sides); none: nothing to walk
__host__ __device__ __forceinline__ GridWalk grid_walk_begin(const GridArgs &ga, f3 o, f3 d, f3 inv, bool wanted) {
    GridWalk w;
    const float gx0 = ga.gmin[0], gy0 = ga.gmin[1], gz0 = ga.gmin[2], hx = ga.h[0], hy = ga.h[1], hz = ga.h[2];
    const int nx = ga.n[0], ny = ga.n[1], nz = ga.n[2];
    const float gx1 = __builtin_fmaf((float)nx, hx, gx0), gy1 = __builtin_fmaf((float)ny, hy, gy0), gz1 = __builtin_fmaf((float)nz, hz, gz0);
    const float ax0 = (gx0 - o.x) * inv.x, ax1 = (gx1 - o.x) * inv.x;
    const float ay0 = (gy0 - o.y) * inv.y, ay1 = (gy1 - o.y) * inv.y;
    const float az0 = (gz0 - o.z) * inv.z, az1 = (gz1 - o.z) * inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.0f));
    const float t1 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    const float t0m = t0 - __builtin_fmaf(fabsf(t0), 2e-5f, 1e-30f), t1p = t1 + __builtin_fmaf(fabsf(t1), 2e-5f, 1e-30f);
    w.walking = wanted && !(t0m > t1p);
    const f3 ps = mk(__builtin_fmaf(d.x, t0, o.x), __builtin_fmaf(d.y, t0, o.y), __builtin_fmaf(d.z, t0, o.z));
    const bool ngx = inv.x < 0.0f, ngy = inv.y < 0.0f, ngz = inv.z < 0.0f;
    // (clamped as floats: a degenerate ray's entry point may be far outside what an int holds; NaN -> 0)
    const int ix = (int)fminf(fmaxf(floorf((ps.x - gx0) * ga.inv_h[0]), 0.0f), (float)(nx - 1));
    const int iy = (int)fminf(fmaxf(floorf((ps.y - gy0) * ga.inv_h[1]), 0.0f), (float)(ny - 1));
    const int iz = (int)fminf(fmaxf(floorf((ps.z - gz0) * ga.inv_h[2]), 0.0f), (float)(nz - 1));
    w.c0 = (float)(ix + nx * (iy + ny * iz));
    w.scx = ngx ? -1.0f : 1.0f; w.scy = (float)(ngy ? -nx : nx); w.scz = (float)(ngz ? -(nx * ny) : nx * ny);
    w.kmx = (float)(ngx ? ix : nx - 1 - ix); w.kmy = (float)(ngy ? iy : ny - 1 - iy); w.kmz = (float)(ngz ? iz : nz - 1 - iz);
    w.kx = 0.0f; w.ky = 0.0f; w.kz = 0.0f;
    // first boundary ahead per axis, then one cell size further per step
    w.t0x = (__builtin_fmaf((float)(ngx ? ix : ix + 1), hx, gx0) - o.x) * inv.x;
    w.t0y = (__builtin_fmaf((float)(ngy ? iy : iy + 1), hy, gy0) - o.y) * inv.y;
    w.t0z = (__builtin_fmaf((float)(ngz ? iz : iz + 1), hz, gz0) - o.z) * inv.z;
    w.dtx = hx * fabsf(inv.x); w.dty = hy * fabsf(inv.y); w.dtz = hz * fabsf(inv.z);
    w.tx = w.t0x; w.ty = w.t0y; w.tz = w.t0z;
    w.selmask = (ngx ? 8u : 1u) | (ngy ? 16u : 2u) | (ngz ? 32u : 4u);
    w.emask = 0x40u;
    return w;
}

__host__ __device__ __forceinline__ uint32_t grid_walk_cell(const GridWalk &w) {
    return (uint32_t)__builtin_fmaf(w.kz, w.scz, __builtin_fmaf(w.ky, w.scy, __builtin_fmaf(w.kx, w.scx, w.c0)));
}

// to the next cell: the axis whose boundary comes first; a tie steps on every tied axis at once (the cells skipped touch
// the ray in a point only, and whatever lies that close to it is listed in the cell entered as well: the margin).
// Branch-free; once the ray has left the grid (`walking` false) the rest of the state is meaningless.
__host__ __device__ __forceinline__ void grid_walk_step(GridWalk &w) {
    const float tm = fminf(fminf(w.tx, w.ty), w.tz);
    const bool sx = w.tx <= tm, sy = w.ty <= tm, sz = w.tz <= tm;
    const bool out = (sx & (w.kx == w.kmx)) | (sy & (w.ky == w.kmy)) | (sz & (w.kz == w.kmz));
    w.kx += sx ? 1.0f : 0.0f; w.ky += sy ? 1.0f : 0.0f; w.kz += sz ? 1.0f : 0.0f;
    w.tx = __builtin_fmaf(w.kx, w.dtx, w.t0x); w.ty = __builtin_fmaf(w.ky, w.dty, w.t0y); w.tz = __builtin_fmaf(w.kz, w.dtz, w.t0z);
    w.emask = w.selmask & ((sx ? 9u : 0u) | (sy ? 18u : 0u) | (sz ? 36u : 0u));
    w.walking = w.walking & !out;
}

// a reference (GridArgs) is new to a ray in a cell entered with `emask`
__host__ __device__ __forceinline__ bool grid_flags_new(uint32_t flags, uint32_t emask) { return (((flags & 0x3Fu) | 0x40u) & emask) != 0u; }
__host__ __device__ __forceinline__ bool grid_ref_is_new(uint32_t ref, uint32_t emask) { return grid_flags_new(ref >> 8, emask); }    // narrow reference

// cells the walk of a ray will visit, estimated: 1 + the cell boundaries its span inside the grid's box crosses, counted as
// the span's extent in cell units per axis (off by at most one and a half cells; 0: the ray misses the box).  k_path_w sorts the
// survivors of a bounce by it, so that the rays of a FRESH group walk about equally far and the walk's trips run on full waves.
__host__ __device__ __forceinline__ uint32_t grid_walk_length(const GridArgs &ga, f3 o, f3 d, f3 inv) {
    const float gx0 = ga.gmin[0], gy0 = ga.gmin[1], gz0 = ga.gmin[2];
    const float gx1 = __builtin_fmaf((float)ga.n[0], ga.h[0], gx0), gy1 = __builtin_fmaf((float)ga.n[1], ga.h[1], gy0), gz1 = __builtin_fmaf((float)ga.n[2], ga.h[2], gz0);
    const float ax0 = (gx0 - o.x) * inv.x, ax1 = (gx1 - o.x) * inv.x;
    const float ay0 = (gy0 - o.y) * inv.y, ay1 = (gy1 - o.y) * inv.y;
    const float az0 = (gz0 - o.z) * inv.z, az1 = (gz1 - o.z) * inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.0f));
    const float t1 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    if (!(t0 <= t1)) return 0u;
    const float per_t = __builtin_fmaf(fabsf(d.z), ga.inv_h[2], __builtin_fmaf(fabsf(d.y), ga.inv_h[1], fabsf(d.x) * ga.inv_h[0]));
    return 1u + (uint32_t)fminf((t1 - t0) * per_t, 1000.0f);
}

// may the ray be walked at all?  finite, a direction of sane length, the origin within `reach` of the grid's centre
__host__ __device__ __forceinline__ bool grid_walk_sane(const GridArgs &ga, f3 o, f3 d) {
    return fabsf(o.x - ga.centre[0]) <= ga.reach && fabsf(o.y - ga.centre[1]) <= ga.reach && fabsf(o.z - ga.centre[2]) <= ga.reach &&
           fabsf(d.x) <= 4.0f && fabsf(d.y) <= 4.0f && fabsf(d.z) <= 4.0f;
}

// ---- camera groups of k_path_w: one cone for the 64 rays, one pass over the primitives' bounds --------------------------
// The camera rays of a FRESH group share their origin (pinhole, with or without jitter) and fan out over a degree or two.  Instead
// of 64 walks through the grid, the wave (lane = one primitive) tests every primitive's bound against the CONE of the group --
// apex = the common origin, axis = the normalised sum of the first and the last ray's direction, half angle = the largest angle any
// of the 64 directions makes with the axis (measured, so the reference's normalize(R) quirk and jitter are covered as they are) --
// and only the few primitives the cone meets have their bounds tested by the rays themselves (wave-uniform index, like the big
// primitives).  The cone test is cull-side arithmetic: it only has to be a superset of what any ray of the group can hit, and it
// is, with margins of 1e-4 of the distances involved (float error: 1e-6) -- checked on the host by tests/test_grid_cpu.py through
// pt_debug_fan_probe, which runs these very functions.
struct FanCone { f3 apex, axis; float cs, sn; f3 nrm; float so; };   // cos / sin of the (loosened) half angle; the fan's plane through the
                                                                    // apex (unit normal) and the sine of the largest angle a ray leaves it by
__host__ __device__ __forceinline__ f3 fan_axis(f3 d_first, f3 d_last) {
    const f3 s = mk(d_first.x + d_last.x, d_first.y + d_last.y, d_first.z + d_last.z);
    const float n2 = __builtin_fmaf(s.z, s.z, __builtin_fmaf(s.y, s.y, s.x * s.x));
    const float inv = 1.0f / __builtin_sqrtf(n2 > 1e-30f ? n2 : 1e-30f);
    return mk(s.x * inv, s.y * inv, s.z * inv);
}
// the plane the group's rays lie in, nearly (a pixel row: the first and the last direction span it; one ray: any plane through it)
__host__ __device__ __forceinline__ f3 fan_normal(f3 d_first, f3 d_last) {
    f3 n = mk(d_first.y * d_last.z - d_first.z * d_last.y, d_first.z * d_last.x - d_first.x * d_last.z, d_first.x * d_last.y - d_first.y * d_last.x);
    float n2 = __builtin_fmaf(n.z, n.z, __builtin_fmaf(n.y, n.y, n.x * n.x));
    if (!(n2 > 1e-12f)) {                                    // parallel directions: a normal of the first one
        const f3 u = fabsf(d_first.x) < 0.6f ? mk(1.0f, 0.0f, 0.0f) : mk(0.0f, 1.0f, 0.0f);
        n = mk(d_first.y * u.z - d_first.z * u.y, d_first.z * u.x - d_first.x * u.z, d_first.x * u.y - d_first.y * u.x);
        n2 = __builtin_fmaf(n.z, n.z, __builtin_fmaf(n.y, n.y, n.x * n.x));
    }
    const float inv = 1.0f / __builtin_sqrtf(n2 > 1e-30f ? n2 : 1e-30f);
    return mk(n.x * inv, n.y * inv, n.z * inv);
}
// `min_dot` = the smallest dot(direction, axis) over the group's rays, `max_off` = the largest |dot(direction, normal)|.
// False: the fan is too wide to be worth a cone (or not finite)
__host__ __device__ __forceinline__ bool fan_finish(f3 apex, f3 axis, float min_dot, f3 normal, float max_off, FanCone &c) {
    c.apex = apex; c.axis = axis; c.nrm = normal;
    c.so = max_off * 1.0001f + 1e-5f;                            // (directions are unit to 1e-6)
    const float cs = min_dot * (1.0f - 1e-5f) - 1e-5f;           // a larger angle
    c.cs = cs;
    c.sn = __builtin_sqrtf(fmaxf(1.0f - cs * cs, 0.0f));
    return cs > 0.995f && cs <= 1.0f;                             // half angle above 5.7 degrees (a group that wraps from one pixel row to the next, a frame a few hundred pixels wide): no cone -- it would meet too much of the scene; (NaN: false)
}
// may a ray of the cone meet the bound?  sphere: centre bmin.xyz, radius bmax[3]; cube: the sphere around its AABB
__host__ __device__ __forceinline__ bool fan_meets(const float *bmin, const float *bmax, bool sphere, const FanCone &c) {
    f3 ctr;
    float r;
    if (sphere) { ctr = mk(bmin[0], bmin[1], bmin[2]); r = bmax[3]; }
    else {
        ctr = mk(0.5f * (bmin[0] + bmax[0]), 0.5f * (bmin[1] + bmax[1]), 0.5f * (bmin[2] + bmax[2]));
        const f3 h = mk(bmax[0] - ctr.x, bmax[1] - ctr.y, bmax[2] - ctr.z);
        r = __builtin_sqrtf(__builtin_fmaf(h.z, h.z, __builtin_fmaf(h.y, h.y, h.x * h.x))) * 1.000001f;
    }
    const f3 v = mk(ctr.x - c.apex.x, ctr.y - c.apex.y, ctr.z - c.apex.z);
    const float n2 = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
    const float s = __builtin_fmaf(v.z, c.axis.z, __builtin_fmaf(v.y, c.axis.y, v.x * c.axis.x));
    const f3 w = mk(__builtin_fmaf(-s, c.axis.x, v.x), __builtin_fmaf(-s, c.axis.y, v.y), __builtin_fmaf(-s, c.axis.z, v.z));   // v minus its axial part: no cancellation in p
    const float p = __builtin_sqrtf(__builtin_fmaf(w.z, w.z, __builtin_fmaf(w.y, w.y, w.x * w.x)));
    const float rr = __builtin_fmaf(1e-4f, r + __builtin_sqrtf(n2), r) + 1e-30f;
    // distance of the centre from the solid cone >= p cos - s sin (equality beside the cone's flank; behind the apex the true distance
    // is |v|, which is larger): outside only if that exceeds the radius.  A false comparison (NaN) keeps the primitive.
    // ... and the rays leave the fan's plane by at most asin(so): a centre farther from the plane than the radius + |v| so is out of
    // every ray's reach (a point of a ray at distance t from the apex is within t so of the plane, and t <= |v| + r where it can touch)
    const float off = fabsf(__builtin_fmaf(v.z, c.nrm.z, __builtin_fmaf(v.y, c.nrm.y, v.x * c.nrm.x)));
    const float reach = __builtin_fmaf(c.so, __builtin_sqrtf(n2) + r, rr);
    return !(__builtin_fmaf(p, c.cs, -(s * c.sn)) > rr) && !(off > reach);
}


__host__ __device__ inline uint32_t p_cursor_offset(int G, int M) { return q_lds_offset(G, M); }
__host__ __device__ inline uint32_t p_queue_offset(int G, int M) { return p_cursor_offset(G, M); }
__host__ __device__ inline uint32_t p_lds_bytes(int G, int M, uint32_t cap) { return p_queue_offset(G, M) + (uint32_t)kWaves * cap * kPFields * 4u; }

// k_path_q<MESH>: rays whose next candidate is a MESH primitive wait on a third typed stack, in the wave's arena behind the
// parked words (field f of entry s at moff + f * kMStack + s): the ray's whole test state -- what a queue record holds, the
// best hit so far (depth, primitive, face, P, N) and the traversal's own state (node cursor, best triangle as a 64-bit key
// t-bits << 32 | index in the file, its position in the blob) -- so that a traversal can be interrupted and resumed.
// A MESH turn pops 64 of them and works the threaded BVHs off in two dense stages: WALK (lane = ray, one node per trip; a
// leaf becomes a (ray lane, first triangle, count) pair in LDS) and TRI (lane = one (ray, triangle) pair: the ray comes
// over from its lane by bpermute, the result goes into the ray's key by an LDS atomic min).
#ifndef PT_MESH_TURN
#define PT_MESH_TURN 128                 // a MESH turn runs when this many rays wait (or nothing else can run): 64 are popped, the
#endif                                   //   others are the reserve that lets a thinned-out WALK stop (kMeshMinWalk)
#ifndef PT_MESH_MIN_WALK
#define PT_MESH_MIN_WALK 16
#endif
constexpr uint32_t kMeshTurn = PT_MESH_TURN;
constexpr uint32_t kMStack = kMeshTurn + 128;   // entries (bound: below kMeshTurn before a push of at most 64; a MESH turn pops 64 and pushes fewer than 64 + 64)
constexpr uint32_t kMFields = 24;        // o d thr pixelword mask candidate|level<<8 | best hit|face P N | node keylo keyhi position
#ifndef PT_MESH_WALK_STEPS
#define PT_MESH_WALK_STEPS 3
#endif
constexpr uint32_t kMeshWalkSteps = PT_MESH_WALK_STEPS;      // BVH nodes a WALK lane visits per trip of the stage dispatcher
constexpr uint32_t kMPairs = 64u + 64u * kMeshWalkSteps;     // (ray, triangle) pairs per wave in LDS: fewer than 64 before a trip adds up to 64 per step
constexpr uint32_t kMScratchBytes = 64u * 8u + 64u * 4u + kMPairs * 4u;     // keys, positions, pairs
constexpr uint32_t kMeshMinWalk = PT_MESH_MIN_WALK;   // a WALK below this many lanes stops while other rays wait: its rays go back with their cursors
constexpr int kMeshPairTris = 1 << 24;   // a pair entry holds lane (6 bits), triangles left in the leaf (2), triangle (24)
__host__ __device__ inline uint32_t p_mesh_offset(int G, int M, uint32_t cap) { return (p_lds_bytes(G, M, cap) + 15u) & ~15u; }
__host__ __device__ inline uint32_t p_mesh_lds_bytes(int G, int M, uint32_t cap) { return p_mesh_offset(G, M, cap) + (uint32_t)kWaves * kMScratchBytes; }


struct FoldArgs {
    float *image;
    float *planes;
    size_t plane_stride;
    uint32_t batch, n_own;
    int W, row_offset, row_stride;
};

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

// ---------------------------------------------------------------- launchers (host side of each family) ----
// *_setup: raise the dynamic-LDS limit of the variant's kernels when `lds_bytes` needs it and report how many
// blocks of it fit on a CU; *_launch: enqueue on `stream` (the caller checks hipGetLastError()).
struct SegVariant { bool geom_lds, cull, nee, wide; };
hipError_t seg_setup(const SegVariant &v, uint32_t lds_bytes, int *blocks_per_cu);
void seg_launch(const SegVariant &v, bool last, bool gen, int grid, uint32_t lds_bytes, hipStream_t stream, const SegArgs &a,
                const GeomRec *geoms, const MatRec *mats);
void generate_launch(hipStream_t stream, const GenArgs &g, uint32_t work);

hipError_t queue_setup(bool mesh, uint32_t lds_bytes, int *blocks_per_cu);
void queue_launch(bool mesh, bool last, bool gen, int grid, uint32_t lds_bytes, hipStream_t stream, const SegArgs &a,
                  const GeomRec *geoms, const MatRec *mats, const QTables &qt);

hipError_t path_setup(bool mesh, bool nee, int cap, uint32_t lds_bytes, int *blocks_per_cu);
void path_launch(bool mesh, bool nee, int cap, int grid, uint32_t lds_bytes, hipStream_t stream, const SegArgs &a, const PathArgs &pa,
                 const GeomRec *geoms, const MatRec *mats, const QTables &qt);
constexpr uint32_t kNeeExtraFields = 9;  // k_path_q<NEE>: words a stack / parked record holds beyond the common ones

// k_path_w (pt_k_wide.hip): `shape` 0 / 1 = narrow ids (33..256 primitives, tables in LDS), 2 / 3 = wide ids (more than 256
// primitives, geometry gathered from global memory); the layout says what a shape needs
struct WideLayout { uint32_t waves_per_block, slots_per_wave, payload_per_wave, stack_slots, lds_bytes; };
uint32_t wide_lds_bytes(int shape, int G, int M, uint32_t grid_bytes);
hipError_t wide_setup(int shape, int G, int M, uint32_t grid_bytes, WideLayout *out);
void wide_launch(int shape, int grid, uint32_t lds_bytes, hipStream_t stream, const SegArgs &a, const PathArgs &pa, const GridArgs &ga,
                 const GeomRec *geoms, const MatRec *mats, const FaceFrame *frames);

void fold_launch(hipStream_t stream, const FoldArgs &f);
void flat_launch(hipStream_t stream, const FlatArgs &f, uint32_t lds_bytes);
void display_launch(hipStream_t stream, const float *image, uchar4 *out, uint32_t n, float scale);
void kat_rng_from_thread(hipStream_t stream, float resx, float time, int n, const int *xy, float *out);
void kat_hemisphere(hipStream_t stream, int n, const float *nrm, const float *xi, float *out);
void kat_light_points(hipStream_t stream, const GeomRec *g, int n, const float *seeds, float *out);
void kat_sincos(hipStream_t stream, int n, const float *a, float *s, float *c);
#ifdef PT_CULL_STATS
void cull_stats_seg(unsigned long long *acc16);      // analysis builds: every family adds its own counters
void cull_stats_queue(unsigned long long *acc16);
void cull_stats_path(unsigned long long *acc16);
void cull_stats_wide(unsigned long long *acc16);
void phase_cycles_wide(unsigned long long *out16);     // k_path_w: cycles per phase, summed over the waves
void stats_wide(unsigned long long *out32);            // k_path_w: stage statistics (indices: pt_k_wide.hip)
#endif

}  // namespace ptk
