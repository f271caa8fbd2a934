// pt_k_wide.hip -- `ordering = 2` for scenes of 33..256 analytic primitives: whole paths in ONE launch per group of
// iterations, with every expensive step on dense, type-pure waves.
//
// The nearest-hit loop it replaces is type- and count-agnostic (/root/reference/src/raytraceKernel.cu:134-153: every
// primitive, first strictly nearer wins, ties to the lower index).  The lock-step kernel for these scenes
// (k_bounce_seg<WIDE>, pt_k_seg.hip) finds the same hits with a two-level culling, but its per-lane loops leave the
// wave mostly idle: the walk over a lane's clusters runs at 24 of 64 lanes, the exact cube tests at 22, the exact
// sphere tests at 7 (profiles/r02i_cullstats_c4.log).  Here the unit of work is no longer "a ray":
//
//   FRESH   64 rays (the top of the wave's stack of survivors, or a job of camera rays) take a RAY SLOT each in the
//           wave's LDS region (origin + direction, an 8-entry candidate list, best depth / hit so far) and park
//           throughput, pixel word and level in the slot's payload record in global memory.  A wave-uniform pass tests
//           the <= 32 + 32 cluster boxes (cube clusters / sphere clusters) and leaves two 32-bit masks per lane.
//   PAIRS   the (ray, cluster) pairs of those 64 rays are EXPANDED: ballot-free prefix sums give every pair its index,
//           the lanes write their pairs into a small LDS buffer (a cheap per-lane loop: ctz, store), and the wave then
//           processes the buffer 64 pairs at a time -- lane = one pair: gather the ray from its slot, test the
//           cluster's members' own bounds (cubes first, then spheres: one code path per group but the boundary one),
//           append every candidate to the ray's list as a 16-bit key (quantised conservative entry distance, id).
//           Full waves whatever a single ray's cluster count is.
//   SELECT  every ray picks its nearest candidate (smallest key) and waits on one of two wave-private stacks of slot
//           ids by that candidate's TYPE.  A ray whose list overflowed (> 8 candidates: rare) takes the reference
//           loop itself on the spot and waits for one confirming test of the winner.
//   TEST    pops 64 slots of one type: the exact reference test of the current candidate on all lanes; the winner so
//           far is kept as (depth, id, face) in the slot.  Then the next candidate that could still win or tie -- key
//           distance not beyond the best hit -- is selected by a branch-free scan of the 8 keys (no re-evaluation of
//           bounds), and the ray goes back on the stack of THAT candidate's type (hit point and normal of a winning
//           test go to the payload record meanwhile).  A ray without such a candidate is finished: shaded at once
//           (RNG stream of its own level, scatter, emitters -> memory-side float atomics), its slot freed, a survivor
//           pushed on the wave's stack in global memory for the next bounce.
//
// Nothing leaves the wave: no barrier, no inter-wave traffic, no pools.  One 1024-thread block per CU shares the
// 37-KB geometry table (variants: fewer waves with more ray slots each).  Results are the reference loop's: every
// primitive whose conservative bound the ray enters before the best hit is tested exactly; image, live counts and
// emitter hits equal every other kernel's bit for bit (tests/test_gpu_parity.py, tests/test_gpu_round3.py).
#include "pt_kernels.hpp"

namespace ptk {

namespace {

// slot fields (SoA, stride R dwords, wave-private LDS)
enum : uint32_t { F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_L0, F_L1, F_L2, F_L3, F_META, F_BEST, F_COUNT };
// payload fields (SoA, stride R floats, global memory, per wave)
enum : uint32_t { P_TX = 0, P_TY, P_TZ, P_PV, P_LEVEL, P_PX, P_PY, P_PZ, P_NX, P_NY, P_NZ };
static_assert(P_NZ + 1 == kWPayload, "payload record");

constexpr uint32_t kMetaHasHit = 1u << 27;
constexpr uint32_t kListCap = 8;

// quantised conservative entry distance: floor(max(tn, 0) * qscale), 0..254 (255 would collide with the empty key 0xFFFF)
__device__ __forceinline__ uint32_t quant_tn(float tn, float qscale) {
    const float v = fminf(fmaxf(tn, 0.0f) * qscale, 254.0f);
    return (uint32_t)v;
}

// the smallest viable key among the 8 entries of a list: key = q << 8 | id, empty = 0xFFFF; viable: q <= qmax.
// Returns the key (0xFFFF: none) and its position.
__device__ __forceinline__ uint32_t select_next(const uint32_t L[4], uint32_t qmax, uint32_t &pos) {
    uint32_t bestk = 0xFFFFu, bestp = 0u;
    const uint32_t limit = (qmax << 8) | 0xFFu;           // keys above it are beyond the best hit (and 0xFFFF > limit: qmax <= 254)
#pragma unroll
    for (uint32_t k = 0; k < kListCap; ++k) {
        const uint32_t key = (k & 1u) ? (L[k >> 1] >> 16) : (L[k >> 1] & 0xFFFFu);
        const bool better = key <= limit && key < bestk;
        bestk = better ? key : bestk;
        bestp = better ? k : bestp;
    }
    pos = bestp;
    return bestk;
}

}  // namespace

template <int WAVES, int R, int PB>
__global__ __launch_bounds__(WAVES * 64) void k_path_w(SegArgs a, PathArgs pa, const GeomRec *__restrict__ geoms,
                                                        const MatRec *__restrict__ mats, const FaceFrame *__restrict__ frames) {
    constexpr uint32_t STK = (uint32_t)((R + 64 + 63) / 64 * 64);      // survivors' stack: at most 63 + R rays wait for their next bounce
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);   // [1] emitted (block sum), [18..31] parked constants, [32..96] survivors per level
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
    stage_tables(smem, geoms, a.G, mats, a.M, true, lg, lm, a.cluster_bytes);        // ends with __syncthreads()
    const ClusterRec *cl = reinterpret_cast<const ClusterRec *>(lg + a.G);
    const int nbc = a.nbc, nsc = a.nsc;
    const unsigned char *members = reinterpret_cast<const unsigned char *>(cl + nbc + nsc);
    const int nmem = nbc + nsc > 0 ? cl[nbc + nsc - 1].first + cl[nbc + nsc - 1].count : 0;      // analytic primitives (MESH objects without data are in no cluster)

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t wslot = blockIdx.x * WAVES + wave;
    const uint32_t nwaves = gridDim.x * WAVES;
    const uint32_t D = pa.depth;
    // the wave's LDS region: 12 slot fields x R | typed stacks of slot ids (cubes up, spheres down) | free list | pair buffer
    uint32_t *wl = reinterpret_cast<uint32_t *>(smem + tables_bytes(a.G, a.M, true) + a.cluster_bytes) + (size_t)wave * (14u * R + PB);
    float *wf = reinterpret_cast<float *>(wl);
    uint32_t *xstack = wl + 12u * R, *freel = wl + 13u * R, *pairbuf = wl + 14u * R;
    for (uint32_t i = lane; i < (uint32_t)R; i += 64u) freel[i] = i;
    uint32_t nfree = R;

    // the wave's arena in global memory: the survivors' stack (kSFields x STK) and the slots' payload (kWPayload x R)
    const uint32_t wave_floats = kSFields * STK + kWPayload * (uint32_t)R;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.arena, 0, pa.arena_bytes, 0x00020000);
    const uint32_t woff = wslot * wave_floats;             // in floats; the arena is below 4 GiB (checked by the host)
    const uint32_t poff = woff + kSFields * STK;
    auto ring_ld = [&](uint32_t off, uint32_t f) -> float { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, f * STK * 4u, 0)); };
    auto ring_st = [&](uint32_t off, uint32_t f, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off * 4u, f * STK * 4u, 0); };
    auto pay_ld = [&](uint32_t sid, uint32_t f) -> float { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (poff + sid) * 4u, f * (uint32_t)R * 4u, 0)); };
    auto pay_st = [&](uint32_t sid, uint32_t f, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (poff + sid) * 4u, f * (uint32_t)R * 4u, 0); };

    uint32_t *bank = a.bank ? a.sync->counts_b : a.sync->counts;
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        uint32_t *other = a.bank ? a.sync->counts : a.sync->counts_b;
        a.sync->totals[threadIdx.x] += other[threadIdx.x];
        other[threadIdx.x] = 0u;
        if (threadIdx.x == 0) bank[0] = a.n_rays;
    }

    uint32_t emitted = 0u;
    uint32_t nbox = 0u, nsph = 0u;                         // slots waiting on the two typed stacks
    uint32_t sp = 0u;                                      // rays on the wave's stack of survivors
    uint32_t jobpos = 0u, jobend = 0u;
    bool tickets_left = true;
    uint32_t next_ticket = 0u, round = 0u;
    uint32_t ctr = wslot % kTicketCtrs, dry = 0u;
    if (pa.static_rounds == 0u && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);
    uint32_t turns = 0u;
    const float kInf = 100000000000000000.0f;
    const float qscale = pa.qscale, slack_max = pa.slack_max;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (;;) {
        if (++turns > pa.turn_limit) { if (lane == 0) *pa.error = 3u; break; }             // never reached; bounds a broken build
        int act;                                                                           // 0 FRESH from the stack, 3 FRESH camera rays, 1 TEST cubes, 2 TEST spheres
        if (nbox >= 64u) act = 1;
        else if (nsph >= 64u) act = 2;
        else if (nfree >= 64u) {
            if (sp >= 64u) act = 0;
            else {
                while (jobpos >= jobend && tickets_left) {                                 // next job of camera rays (a dry counter: try the next)
                    unsigned long long job;
                    if (round < pa.static_rounds) {
                        job = (unsigned long long)wslot * pa.static_rounds + round;       // the wave's own contiguous range
                        round++;
                        if (round == pa.static_rounds && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);
                    } else {
                        job = (unsigned long long)pa.static_rounds * nwaves + (unsigned long long)__builtin_amdgcn_readfirstlane(next_ticket) * kTicketCtrs + ctr;
                        if (job * pa.job_rays >= (unsigned long long)a.n_rays) {
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
        } else act = nbox >= nsph ? 1 : 2;                                                 // no room for a fresh group: the fuller stack pops what it has

        if (act == 0 || act == 3) {
            // ================================================================ FRESH
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
            } else {                                                                       // the top of the wave's stack of survivors
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
#ifdef PT_CULL_STATS
            qstat(0, 1ull); qstat(1, (unsigned long long)__popcll(__ballot(valid)));
#endif
            // a slot for every ray of the group
            const u64 vb = __ballot(valid);
            const uint32_t nv = (uint32_t)__popcll(vb);
            uint32_t sid = 0u;
            if (valid) {
                sid = freel[nfree - 1u - wave_rank(vb)];
                wf[F_OX * R + sid] = o.x; wf[F_OY * R + sid] = o.y; wf[F_OZ * R + sid] = o.z;
                wf[F_DX * R + sid] = d.x; wf[F_DY * R + sid] = d.y; wf[F_DZ * R + sid] = d.z;
                wl[F_L0 * R + sid] = 0xFFFFFFFFu; wl[F_L1 * R + sid] = 0xFFFFFFFFu; wl[F_L2 * R + sid] = 0xFFFFFFFFu; wl[F_L3 * R + sid] = 0xFFFFFFFFu;
                wl[F_META * R + sid] = 0u;                                                 // candidate count while the pairs are expanded
                pay_st(sid, P_TX, thr.x); pay_st(sid, P_TY, thr.y); pay_st(sid, P_TZ, thr.z);
                pay_st(sid, P_PV, __uint_as_float(pv)); pay_st(sid, P_LEVEL, __uint_as_float(level));
            }
            nfree -= nv;
            // ---------------------------------------------------------------- wave-uniform pass over the cluster boxes
            const CullRay cr = make_cull_ray(o, d);
            uint32_t mB = 0u, mS = 0u;
            for (int c = 0; c < nbc; ++c) {
                float tn;
                if (cull_box(cl[c].bmin, cl[c].bmax, cr, tn)) mB |= 1u << c;
            }
            for (int c = 0; c < nsc; ++c) {
                float tn;
                if (cull_box(cl[nbc + c].bmin, cl[nbc + c].bmax, cr, tn)) mS |= 1u << c;
            }
            if (!valid) { mB = 0u; mS = 0u; }
            // ---------------------------------------------------------------- (ray, cluster) pairs, dense
            // pair index: all cube pairs in lane order, then all sphere pairs (exclusive prefix sums over the wave)
            uint32_t cB = (uint32_t)__popc(mB), cS = (uint32_t)__popc(mS);
            uint32_t nextB = cB, nextS = cS;
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {                                       // inclusive scans
                const uint32_t tb = __shfl_up(nextB, sft), ts = __shfl_up(nextS, sft);
                if (lane >= (uint32_t)sft) { nextB += tb; nextS += ts; }
            }
            const uint32_t TB = __builtin_amdgcn_readlane(nextB, 63), TS = __builtin_amdgcn_readlane(nextS, 63);
            nextB -= cB; nextS = nextS - cS + TB;                                          // exclusive; spheres follow the cubes
            const uint32_t T = TB + TS;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                         // the slots are written before any pair lane reads them
            __builtin_amdgcn_wave_barrier();
            for (uint32_t base = 0u; base < T; base += (uint32_t)PB) {
                const uint32_t end = base + (uint32_t)PB;
                while (mB != 0u && nextB < end) {                                          // cheap per-lane loops: one store per pair
                    const uint32_t b = (uint32_t)__builtin_ctz(mB);
                    mB &= mB - 1u;
                    pairbuf[nextB - base] = (sid << 8) | b;
                    nextB++;
                }
                while (mS != 0u && nextS < end) {
                    const uint32_t b = (uint32_t)__builtin_ctz(mS);
                    mS &= mS - 1u;
                    pairbuf[nextS - base] = (sid << 8) | ((uint32_t)nbc + b);
                    nextS++;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t here = T - base < (uint32_t)PB ? T - base : (uint32_t)PB;
                for (uint32_t c0 = 0u; c0 < here; c0 += 64u) {
                    const uint32_t idx = base + c0 + lane;
                    const bool pvalid = c0 + lane < here;
                    const bool isb = idx < TB;
                    uint32_t psid = 0u, pc = 0u;
                    f3 po = mk(0, 0, 0), pd = mk(0, 0, 1);
                    int first = 0, count = 0;
                    if (pvalid) {
                        const uint32_t e = pairbuf[c0 + lane];
                        psid = e >> 8; pc = e & 0xFFu;
                        po = mk(wf[F_OX * R + psid], wf[F_OY * R + psid], wf[F_OZ * R + psid]);
                        pd = mk(wf[F_DX * R + psid], wf[F_DY * R + psid], wf[F_DZ * R + psid]);
                        first = cl[pc].first; count = cl[pc].count;
                    }
#ifdef PT_CULL_STATS
                    qstat(2, 1ull); qstat(3, (unsigned long long)__popcll(__ballot(pvalid)));
                    { unsigned long long mt = (unsigned long long)count; for (int sft = 32; sft > 0; sft >>= 1) mt += __shfl_down(mt, sft); qstat(13, mt); }
                    { int mx = count; for (int sft = 32; sft > 0; sft >>= 1) { const int t = __shfl_down(mx, sft); mx = t > mx ? t : mx; } qstat(15, (unsigned long long)mx); }
#endif
                    const CullRay pr = make_cull_ray(po, pd);
                    // members' own bounds: per-lane gather from the geometry table; candidates -> the ray's list
                    auto append = [&](uint32_t p, float tn) {
                        const uint32_t pos = atomicAdd(&wl[F_META * R + psid], 1u);
#ifdef PT_CULL_STATS
                        atomicAdd(&g_cull_stats[10], 1ull);
#endif
                        if (pos < kListCap) {
                            unsigned short *l16 = reinterpret_cast<unsigned short *>(&wl[(F_L0 + (pos >> 1)) * R + psid]) + (pos & 1u);
                            *l16 = (unsigned short)((quant_tn(tn, qscale) << 8) | p);
                        }
                    };
                    if (__any(pvalid && isb)) {
                        if (pvalid && isb) {
#pragma unroll 1
                            for (int k = 0; k < count; ++k) {
                                const uint32_t p = members[first + k];
                                const GeomRec *g = lg + p;
                                float tn;
                                if (cull_box(g->bmin, g->bmax, pr, tn)) append(p, tn);
                            }
                        }
                    }
                    if (__any(pvalid && !isb)) {
                        if (pvalid && !isb) {
#pragma unroll 1
                            for (int k = 0; k < count; ++k) {
                                const uint32_t p = members[first + k];
                                const GeomRec *g = lg + p;
                                float tn;
                                if (cull_sphere(g->bmin, g->bmax, pr, tn)) append(p, tn);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                     // the pair buffer is free again, the lists are written
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---------------------------------------------------------------- SELECT: nearest candidate first
            bool queued = false, tobox = false;
            const uint32_t ncand = valid ? wl[F_META * R + sid] : 0u;
            // Rays whose list overflowed (rare: 2 in 10 000 on configs[3]) take the reference loop itself; its winner is then
            // confirmed by one exact test in a TEST group.  A few of them: one ray at a time on the WHOLE wave, every lane
            // testing G / 64 primitives (cubes first, in the order of the member table) + a min-reduction with the
            // reference's tie rule -- a lane alone would keep the wave waiting for G exact tests.  Many: the per-lane loop.
            int ovhit = -1;
            {
                const bool ov = ncand > kListCap;
                const u64 ovb = __ballot(ov);
#ifdef PT_CULL_STATS
                qstat(11, (unsigned long long)__popcll(ovb));
#endif
                if (ovb) {
                    if (__popcll(ovb) >= 12) {
                        if (ov) { float tb; f3 P, N; ovhit = nearest_hit(lg, a.G, o, d, tb, P, N); }
                    } else {
                        u64 m = ovb;
                        while (m) {
                            const int src = (int)__builtin_ctzll(m);
                            m &= m - 1ull;
                            const f3 oo = mk(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                            const f3 dd = mk(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                            float bd = kInf;
                            int bh = -1;
                            for (int k = (int)lane; k < nmem; k += 64) {
                                const int p = (int)members[k];               // every analytic primitive once: cubes, then spheres
                                const GeomRec *g = lg + p;
                                const bool pb = g->type == 1;
                                float depth = -1.0f;
                                f3 P, N;
                                if (__any(pb)) { if (pb) depth = box_test(g->inv, g->xf, g->inside_hits, oo, dd, P, N); }
                                if (__any(!pb)) { if (!pb) depth = sphere_test(g->inv, g->xf, oo, dd, P, N); }
                                if (depth > -PT_EPSILON && (depth < bd || (depth == bd && p < bh))) { bd = depth; bh = p; }
                            }
#pragma unroll
                            for (int sft = 32; sft > 0; sft >>= 1) {             // nearest wins, ties to the lower index
                                const float od = __shfl_xor(bd, sft);
                                const int oh = __shfl_xor(bh, sft);
                                if (oh >= 0 && (bh < 0 || od < bd || (od == bd && oh < bh))) { bd = od; bh = oh; }
                            }
                            if ((int)lane == src) ovhit = bh;
                        }
                    }
                }
            }
            if (valid) {
                const uint32_t cnt = ncand;
                uint32_t first_id = 0u;
                if (cnt > kListCap) {
                    if (ovhit >= 0) {
                        first_id = (uint32_t)ovhit; queued = true;
                        wl[F_L0 * R + sid] = 0xFFFFFFFFu; wl[F_L1 * R + sid] = 0xFFFFFFFFu; wl[F_L2 * R + sid] = 0xFFFFFFFFu; wl[F_L3 * R + sid] = 0xFFFFFFFFu;
                    }
                } else if (cnt != 0u) {
                    uint32_t L[4] = {wl[F_L0 * R + sid], wl[F_L1 * R + sid], wl[F_L2 * R + sid], wl[F_L3 * R + sid]};
                    uint32_t pos;
                    const uint32_t key = select_next(L, 254u, pos);
                    first_id = key & 0xFFu; queued = true;
                    const uint32_t clr = (pos & 1u) ? 0xFFFF0000u : 0x0000FFFFu;           // the chosen entry leaves the list
                    wl[(F_L0 + (pos >> 1)) * R + sid] = L[pos >> 1] | clr;
                }
                if (queued) {
                    wl[F_META * R + sid] = first_id;                                       // current candidate, no hit yet
                    wf[F_BEST * R + sid] = kInf;
                    tobox = lg[first_id].type == 1;
                }
            }
            {
                const u64 bb = __ballot(queued && tobox), sb = __ballot(queued && !tobox), fb = __ballot(valid && !queued);
                if (queued) {
                    const uint32_t pos = tobox ? nbox + wave_rank(bb) : (uint32_t)R - 1u - (nsph + wave_rank(sb));
                    xstack[pos] = sid;
                }
                if (valid && !queued) freel[nfree + wave_rank(fb)] = sid;                  // no candidate at all: the ray leaves the scene
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                nfree += (uint32_t)__popcll(fb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }

        // ==================================================================== TEST (one type per group, any levels)
        const bool isb = act == 1;
        const uint32_t have = isb ? nbox : nsph;
        const uint32_t cnt = have < 64u ? have : 64u;
        const bool valid = lane < cnt;
        const uint32_t qpos = isb ? (have - cnt + lane) : ((uint32_t)R - 1u - (have - cnt + lane));
        if (isb) nbox -= cnt; else nsph -= cnt;
#ifdef PT_CULL_STATS
        qstat(isb ? 4 : 6, 1ull); qstat(isb ? 5 : 7, (unsigned long long)cnt);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                // payload stores of earlier groups have landed ...
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                // ... before they are read back through the same L1
        uint32_t sid = 0u, meta = 0u;
        uint32_t L[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(0, 0, 0);
        float best = kInf;
        uint32_t pv = 0u, level = 0u;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 0);                                  // hit point / normal of the best hit so far
        if (valid) {
            sid = xstack[qpos];
            o = mk(wf[F_OX * R + sid], wf[F_OY * R + sid], wf[F_OZ * R + sid]);
            d = mk(wf[F_DX * R + sid], wf[F_DY * R + sid], wf[F_DZ * R + sid]);
            meta = wl[F_META * R + sid];
            best = wf[F_BEST * R + sid];
            L[0] = wl[F_L0 * R + sid]; L[1] = wl[F_L1 * R + sid]; L[2] = wl[F_L2 * R + sid]; L[3] = wl[F_L3 * R + sid];
            // throughput, pixel word and level: requested now, used after the test
            thr = mk(pay_ld(sid, P_TX), pay_ld(sid, P_TY), pay_ld(sid, P_TZ));
            pv = __float_as_uint(pay_ld(sid, P_PV));
            level = __float_as_uint(pay_ld(sid, P_LEVEL));
            if (meta & kMetaHasHit) {                                         // an earlier test of this ray holds the best hit so far
                P = mk(pay_ld(sid, P_PX), pay_ld(sid, P_PY), pay_ld(sid, P_PZ));
                N = mk(pay_ld(sid, P_NX), pay_ld(sid, P_NY), pay_ld(sid, P_NZ));
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t j = meta & 0xFFu;
        bool has_hit = (meta & kMetaHasHit) != 0u;
        uint32_t hit = (meta >> 8) & 0xFFu;
        int face = (int)((meta >> 16) & 7u) - 1;
        bool won = false;
        {
            const GeomRec *gr = lg + j;                                       // per-lane gather from the LDS table
            float depth = -1.0f;
            f3 p = mk(0, 0, 0), n = mk(0, 0, 0);
            int fc = -1;
            if (valid) {
                if (isb) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, p, fc);
                else depth = sphere_test(gr->inv, gr->xf, o, d, p, n);
            }
            // nearest-hit update of the reference loop: first strictly nearer wins, ties to the lower index
            won = valid && depth > -PT_EPSILON && (has_hit ? (depth < best || (depth == best && j < hit)) : depth < kInf);
            if (won) { best = depth; hit = j; P = p; N = n; face = fc; has_hit = true; }
        }
        // the next candidate that could still win or tie: key distance not beyond the best hit (conservative: one step of slack)
        uint32_t npos = 0u;
        uint32_t qmax = 254u;
        if (has_hit) {
            const float lim = fminf((best + slack_max) * qscale, 253.0f);
            qmax = (uint32_t)lim + 1u;
        }
        const uint32_t nkey = valid ? select_next(L, qmax, npos) : 0xFFFFu;
        const bool more = nkey != 0xFFFFu;
        const bool done = valid && !more;
        bool nbx = false;
        if (more) {
            const uint32_t nid = nkey & 0xFFu;
            const uint32_t clr = (npos & 1u) ? 0xFFFF0000u : 0x0000FFFFu;
            wl[(F_L0 + (npos >> 1)) * R + sid] = L[npos >> 1] | clr;
            wl[F_META * R + sid] = nid | (hit << 8) | ((uint32_t)(face + 1) << 16) | (has_hit ? kMetaHasHit : 0u);
            wf[F_BEST * R + sid] = best;
            if (won) {                                                        // the new best hit's point and normal wait in the payload record
                pay_st(sid, P_PX, P.x); pay_st(sid, P_PY, P.y); pay_st(sid, P_PZ, P.z);
                pay_st(sid, P_NX, N.x); pay_st(sid, P_NY, N.y); pay_st(sid, P_NZ, N.z);
            }
            nbx = lg[nid].type == 1;
        }
#ifdef PT_CULL_STATS
        qstat(8, (unsigned long long)__popcll(__ballot(done && has_hit))); qstat(9, (unsigned long long)__popcll(__ballot(done)));
        qstat(12, (unsigned long long)__popcll(__ballot(more))); qstat(14, (unsigned long long)__popcll(__ballot(more && won)));
#endif
        // ---------------------------------------------------------------- shade the finished rays that hit something
        bool alive = false;
        if (done && has_hit) {
            const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
            const MatRec m = lm[lg[hit].mat];
            if (level + 1u >= D && !(m.emittance > 0.0f)) {
                alive = true;                                                 // depth exhausted: alive, contributes 0
            } else {
                const uint32_t iteration = a.iteration + slot;
                uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + level));
                st = lcg_next(st); const float u_sel = u01(st);
                st = lcg_next(st); const float xi1 = u01(st);
                st = lcg_next(st); const float xi2 = u01(st);
                f3 Lr = mk(0.0f, 0.0f, 0.0f);
                int code = 4;
                const bool hb = lg[hit].type == 1;
                if (__any(hb)) { if (hb) code = scatter_box(m, P, face, frames + 3 * hit, u_sel, xi1, xi2, o, d, thr, Lr); }
                if (__any(!hb)) { if (!hb) code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, Lr); }
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
                    (void)unsafeAtomicAdd(px, Lr.x); (void)unsafeAtomicAdd(px + 1, Lr.y); (void)unsafeAtomicAdd(px + 2, Lr.z);
                    emitted++;
                }
                alive = code <= 2;
            }
        }
        if (alive) atomicAdd(&lsurv[level + 1u], 1u);
        // ---------------------------------------------------------------- requeue / free / survivors
        {
            const u64 bb = __ballot(more && nbx), sb = __ballot(more && !nbx), fb = __ballot(done);
            if (more) {
                const uint32_t pos = nbx ? nbox + wave_rank(bb) : (uint32_t)R - 1u - (nsph + wave_rank(sb));
                xstack[pos] = sid;
            }
            if (done) freel[nfree + wave_rank(fb)] = sid;
            nbox += (uint32_t)__popcll(bb);
            nsph += (uint32_t)__popcll(sb);
            nfree += (uint32_t)__popcll(fb);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool onward = alive && level + 1u < D;
        const u64 ob = __ballot(onward);
        if (ob) {
            const uint32_t n = (uint32_t)__popcll(ob);
            if (sp + n > STK) { if (lane == 0) *pa.error = 2u; }          // never: at most 63 + R rays can wait
            else {
                if (onward) {
                    const uint32_t off = woff + sp + wave_rank(ob);
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

// ------------------------------------------------------------------ host side ---------
namespace {
struct WideShape { int waves, slots, pairbuf; };
// variant -> block shape.  One block per CU; the waves share the CU's LDS: tables + waves x (14 x slots + pairbuf) dwords
constexpr WideShape kShapes[3] = {{16, 120, 128}, {12, 168, 128}, {8, 256, 256}};

template <int WAVES, int R, int PB>
const void *wide_fn() { return reinterpret_cast<const void *>(&k_path_w<WAVES, R, PB>); }
const void *wide_fn_of(int v) {
    return v == 1 ? wide_fn<12, 168, 128>() : v == 2 ? wide_fn<8, 256, 256>() : wide_fn<16, 120, 128>();
}
int clamp_variant(int v) { return v < 0 || v > 2 ? 0 : v; }
}  // namespace

hipError_t wide_setup(int variant, int G, int M, uint32_t cluster_bytes, WideLayout *out) {
    const int v = clamp_variant(variant);
    const WideShape s = kShapes[v];
    const uint32_t lds = tables_bytes(G, M, true) + cluster_bytes + (uint32_t)s.waves * (14u * (uint32_t)s.slots + (uint32_t)s.pairbuf) * 4u;
    out->waves_per_block = (uint32_t)s.waves;
    out->slots_per_wave = (uint32_t)s.slots;
    out->stack_slots = (uint32_t)((s.slots + 64 + 63) / 64 * 64);
    out->lds_bytes = lds;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;        // the tables leave no room for this shape (the caller falls back)
    return hipFuncSetAttribute(wide_fn_of(v), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void wide_launch(int variant, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const PathArgs &pa,
                 const GeomRec *g, const MatRec *m, const FaceFrame *frames) {
    switch (clamp_variant(variant)) {
    case 1: hipLaunchKernelGGL((k_path_w<12, 168, 128>), dim3(grid), dim3(12 * 64), lds, st, a, pa, g, m, frames); break;
    case 2: hipLaunchKernelGGL((k_path_w<8, 256, 256>), dim3(grid), dim3(8 * 64), lds, st, a, pa, g, m, frames); break;
    default: hipLaunchKernelGGL((k_path_w<16, 120, 128>), dim3(grid), dim3(16 * 64), lds, st, a, pa, g, m, frames); break;
    }
}

#ifdef PT_CULL_STATS
void cull_stats_wide(unsigned long long *acc16) {
    unsigned long long v[16];
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_cull_stats), sizeof v) == hipSuccess) for (int i = 0; i < 16; ++i) acc16[i] += v[i];
}
#endif

}  // namespace ptk
