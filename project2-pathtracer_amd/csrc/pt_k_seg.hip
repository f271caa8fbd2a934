// pt_k_seg.hip -- the stable-order bounce kernel (library default) and its variants; k_generate for the parity hook.
// Shared pieces (argument blocks, culling, nearest-hit loops, shading): pt_kernels.hpp.
#include "pt_kernels.hpp"

namespace ptk {

// ------------------------------------------------------------------ generate -----------
__global__ __launch_bounds__(kBlock) void k_generate(GenArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.seg_cnt0 && gid < a.nseg) {
        const uint32_t first = gid * a.seg_slots;
        a.seg_cnt0[gid] = first >= a.n_own ? 0u : (a.n_own - first < a.seg_slots ? a.n_own - first : a.seg_slots);
    }
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        const uint32_t k = threadIdx.x;
        a.sync->totals[k] += a.sync->counts[k];          // fold the previous iteration (stats)
        a.sync->counts[k] = (k == 0) ? a.n_own : 0u;
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

// ------------------------------------------------------------------ bounce, segmented ---
// Wave-autonomous segmented compaction (the default).  The pool is cut into fixed segments of
// S = 64*rpt slots; segment s holds cnt_in[s] live rays packed at its start, in generation
// order.  ONE WAVE owns a segment for the whole launch: it streams the segment 64 rays at a
// time, and survivors go straight from registers to the same segment of the output pool at
// base + running + mbcnt(ballot) -- no inter-wave traffic, no barrier, no ticket, no look-back.
// Global order is still generation order (segments are ordered, each is dense), so the stream
// stays coherent.
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
        // input: the dense prefix of one segment; output: a dense prefix of the same segment of the other pool
        uint32_t n;
        if (GEN) {                                        // level-0 segments are full except the last
            const uint32_t f0 = seg * S;
            n = f0 >= a.n_rays ? 0u : (a.n_rays - f0 < S ? a.n_rays - f0 : S);
        } else {
            n = a.cnt_in[seg];
        }
        const uint32_t base = seg * S;
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
                    const uint32_t gid = base + k;
                    slot = a.batch > 1u ? gid / a.n_own : 0u;
                    const uint32_t local = gid - slot * a.n_own;
                    const uint32_t W = (uint32_t)a.cam.W;
                    const uint32_t lr = local / W, x = local - lr * W;
                    pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
                    camera_ray(a.cam, pixel, a.iteration + slot, o, d);
                    thr = mk(1.0f, 1.0f, 1.0f);
                } else {
                    // wave-uniform field bases (SGPR) + one 32-bit lane offset: `global_load_dword v, v_off, s[base]`
                    const uint32_t idx = base + k;
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

// ------------------------------------------------------------------ host side ---------
namespace {

template <bool LDS, bool LAST, bool CULL, bool GEN, bool NEE = false, bool WIDE = false>
const void *seg_fn() { return reinterpret_cast<const void *>(&k_bounce_seg<LDS, LAST, CULL, GEN, NEE, WIDE>); }

// the four (LAST, GEN) instances of one variant
void seg_fns(const SegVariant &v, const void *out[4]) {
    if (v.nee) { out[0] = seg_fn<true, false, true, false, true>(); out[1] = seg_fn<true, true, true, false, true>(); out[2] = seg_fn<true, false, true, true, true>(); out[3] = seg_fn<true, true, true, true, true>(); }
    else if (v.wide) { out[0] = seg_fn<true, false, true, false, false, true>(); out[1] = seg_fn<true, true, true, false, false, true>(); out[2] = seg_fn<true, false, true, true, false, true>(); out[3] = seg_fn<true, true, true, true, false, true>(); }
    else if (v.geom_lds && v.cull) { out[0] = seg_fn<true, false, true, false>(); out[1] = seg_fn<true, true, true, false>(); out[2] = seg_fn<true, false, true, true>(); out[3] = seg_fn<true, true, true, true>(); }
    else if (v.geom_lds) { out[0] = seg_fn<true, false, false, false>(); out[1] = seg_fn<true, true, false, false>(); out[2] = seg_fn<true, false, false, true>(); out[3] = seg_fn<true, true, false, true>(); }
    else if (v.cull) { out[0] = seg_fn<false, false, true, false>(); out[1] = seg_fn<false, true, true, false>(); out[2] = seg_fn<false, false, true, true>(); out[3] = seg_fn<false, true, true, true>(); }
    else { out[0] = seg_fn<false, false, false, false>(); out[1] = seg_fn<false, true, false, false>(); out[2] = seg_fn<false, false, false, true>(); out[3] = seg_fn<false, true, false, true>(); }
}

template <bool LDS, bool CULL, bool NEE, bool WIDE>
void seg_launch_v(bool last, bool gen, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const GeomRec *g, const MatRec *m) {
    if (gen) {
        if (last) hipLaunchKernelGGL((k_bounce_seg<LDS, true, CULL, true, NEE, WIDE>), dim3(grid), dim3(kBlock), lds, st, a, g, m);
        else hipLaunchKernelGGL((k_bounce_seg<LDS, false, CULL, true, NEE, WIDE>), dim3(grid), dim3(kBlock), lds, st, a, g, m);
    } else {
        if (last) hipLaunchKernelGGL((k_bounce_seg<LDS, true, CULL, false, NEE, WIDE>), dim3(grid), dim3(kBlock), lds, st, a, g, m);
        else hipLaunchKernelGGL((k_bounce_seg<LDS, false, CULL, false, NEE, WIDE>), dim3(grid), dim3(kBlock), lds, st, a, g, m);
    }
}

}  // namespace

hipError_t seg_setup(const SegVariant &v, uint32_t lds_bytes, int *blocks_per_cu) {
    const void *fns[4];
    seg_fns(v, fns);
    if (lds_bytes > 64u * 1024u)
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fns[0], kBlock, lds_bytes) != hipSuccess || occ < 1) occ = 2;
    *blocks_per_cu = occ;
    return hipSuccess;
}

void seg_launch(const SegVariant &v, bool last, bool gen, int grid, uint32_t lds, hipStream_t st, const SegArgs &a,
                const GeomRec *g, const MatRec *m) {
    if (v.nee) seg_launch_v<true, true, true, false>(last, gen, grid, lds, st, a, g, m);
    else if (v.wide) seg_launch_v<true, true, false, true>(last, gen, grid, lds, st, a, g, m);
    else if (v.geom_lds && v.cull) seg_launch_v<true, true, false, false>(last, gen, grid, lds, st, a, g, m);
    else if (v.geom_lds) seg_launch_v<true, false, false, false>(last, gen, grid, lds, st, a, g, m);
    else if (v.cull) seg_launch_v<false, true, false, false>(last, gen, grid, lds, st, a, g, m);
    else seg_launch_v<false, false, false, false>(last, gen, grid, lds, st, a, g, m);
}

void generate_launch(hipStream_t stream, const GenArgs &g, uint32_t work) {
    hipLaunchKernelGGL(k_generate, dim3((work + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, g);
}

#ifdef PT_CULL_STATS
void cull_stats_seg(unsigned long long *acc16) {
    unsigned long long v[16];
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_cull_stats), sizeof v) == hipSuccess) for (int i = 0; i < 16; ++i) acc16[i] += v[i];
}
#endif

}  // namespace ptk
