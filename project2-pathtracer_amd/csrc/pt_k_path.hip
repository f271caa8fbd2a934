// pt_k_path.hip -- `ordering = 2`, <= 32 primitives: whole paths on the typed work queues, ONE launch per group of
// iterations (what bench.py, the adaptor and ptrender run).  Description next to PathArgs in pt_kernels.hpp.
#include "pt_kernels.hpp"

namespace ptk {

// MESH: the scene holds MESH primitives with triangles (they share the spheres' stack; the traversal needs registers the
// common variant must not pay for).
// NEE: direct_light (oracle path_bounce / direct_light).  A path is ONE chain of records: at a diffuse hit
// the shadow ray to the sampled emitter is traced FIRST -- a record like any other, marked, that carries the scattered direction and
// the geometric factors of the sample in its parked words -- and only when it has been resolved (its nearest hit decides whether
// the emitter's radiance is added) does the scattered ray itself go on.  The radiance a path gathers rides with it (`acc`, the
// oracle's L: emitter hits and shadow-ray contributions summed in bounce order from zero) and is added to the path's pixel of its
// iteration's accumulator plane once, where the path ends -- the same additions in the same order as the per-bounce kernels.
template <bool MESH, bool NEE, int CAP>
__global__ __launch_bounds__(kBlock, (MESH || NEE) ? 5 : 6) void k_path_q(SegArgs a, PathArgs pa, const GeomRec *__restrict__ geoms,
                                                      const MatRec *__restrict__ mats, QTables qt) {
    // stack / parked fields of the NEE variant beyond the common ones: acc (3), scattered direction (3), cos at the surface,
    // 1 / pdf of the light sample, squared distance to it.  Level word: level | shadow << 8 | count-emission << 9 | light << 10
    constexpr uint32_t SF = NEE ? kSFields + 9u : kSFields;
    constexpr uint32_t PF = NEE ? kPParked + 9u : kPParked;
    constexpr uint32_t kPCap = (uint32_t)CAP;               // records per wave (pt_kernels.hpp, kPCaps)
    constexpr uint32_t PX = kPParked;                      // where the NEE words start among a record's parked words
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem);   // [1] emitted (block sum), [32..96] survivors per level
    uint32_t *lsurv = ctrl + 32;
    if (threadIdx.x < 2) ctrl[threadIdx.x] = 0u;
    if (threadIdx.x < 65) lsurv[threadIdx.x] = 0u;
    uint32_t *park = ctrl + 18;
    if (threadIdx.x == 0) {
        const unsigned long long pl = (unsigned long long)(uintptr_t)((NEE || a.batch > 1u) ? a.planes : a.image), st = (unsigned long long)a.plane_stride;
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
    // the wave's arena: its stack (field f of slot s at woff + f * KS + s), then what the queue records park (field f of record r
    // at poff + f * kPCap + r)
    // (MESH: the rays on the mesh stack -- up to kMStack -- are part of the population that may end up on the stack of survivors)
    constexpr uint32_t KS = MESH ? kStack + kMStack : kStack;
    const uint32_t wave_floats = SF * KS + PF * kPCap + (MESH ? kMFields * kMStack : 0u);
    const bool ub = pa.arena_bytes != 0u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.arena, 0, pa.arena_bytes, 0x00020000);
    const uint32_t woff = wslot * wave_floats;             // in floats (buffer path: arena below 4 GiB)
    auto ring_ld = [&](uint32_t off, uint32_t f) -> float {         // off = float index of field 0 of the slot
        return ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, f * KS * 4u, 0)) : pa.arena[(size_t)off + f * KS];
    };
    auto ring_st = [&](uint32_t off, uint32_t f, float v) {
        if (ub) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off * 4u, f * KS * 4u, 0);
        else pa.arena[(size_t)off + f * KS] = v;
    };
    const uint32_t poff = woff + SF * KS;
    auto park_ld = [&](uint32_t pos, uint32_t f) -> float {
        return ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (poff + pos) * 4u, f * kPCap * 4u, 0)) : pa.arena[(size_t)poff + pos + f * kPCap];
    };
    auto park_st = [&](uint32_t pos, uint32_t f, float v) {
        if (ub) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (poff + pos) * 4u, f * kPCap * 4u, 0);
        else pa.arena[(size_t)poff + pos + f * kPCap] = v;
    };

    // MESH: the third typed stack (pt_kernels.hpp, kMStack) and the scratch of the mesh stages
    const uint32_t moff = poff + PF * kPCap;
    auto mesh_ld = [&](uint32_t off, uint32_t f) -> float {
        return ub ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, f * kMStack * 4u, 0)) : pa.arena[(size_t)off + f * kMStack];
    };
    auto mesh_st = [&](uint32_t off, uint32_t f, float v) {
        if (ub) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off * 4u, f * kMStack * 4u, 0);
        else pa.arena[(size_t)off + f * kMStack] = v;
    };
    unsigned long long *mkey = reinterpret_cast<unsigned long long *>(smem + p_mesh_offset(a.G, a.M, kPCap) + (MESH ? wave * kMScratchBytes : 0u));
    uint32_t *mposn = reinterpret_cast<uint32_t *>(mkey + 64);
    uint32_t *mpair = mposn + 64;
    constexpr unsigned long long kMeshNoHit = (0x7F61B1E6ull << 32) | 0x7FFFFFFFull;      // (3.0e38f, no triangle): mesh_test's initial best
    uint32_t nmesh = 0u;
    // a ray goes on the mesh stack: `c` lanes, everything a later MESH turn needs
    auto mesh_push = [&](bool c, f3 o, f3 d, f3 thr, uint32_t pv, uint32_t mask, uint32_t jl, float best, int hit, int face, f3 P, f3 N,
                         int node, unsigned long long key, uint32_t mpos) {
        const u64 b = __ballot(c);
        if (b == 0ull) return;
        const uint32_t n = (uint32_t)__popcll(b);
        if (nmesh + n > kMStack) { if (lane == 0) *pa.error = 2u; return; }               // never: see kMStack
        if (c) {
            const uint32_t off = moff + nmesh + wave_rank(b);
            mesh_st(off, 0, o.x); mesh_st(off, 1, o.y); mesh_st(off, 2, o.z);
            mesh_st(off, 3, d.x); mesh_st(off, 4, d.y); mesh_st(off, 5, d.z);
            mesh_st(off, 6, thr.x); mesh_st(off, 7, thr.y); mesh_st(off, 8, thr.z);
            mesh_st(off, 9, __uint_as_float(pv)); mesh_st(off, 10, __uint_as_float(mask)); mesh_st(off, 11, __uint_as_float(jl));
            mesh_st(off, 12, best); mesh_st(off, 13, __uint_as_float((uint32_t)(hit + 1) | ((uint32_t)(face + 1) << 8)));
            if (hit >= 0) {
                mesh_st(off, 14, P.x); mesh_st(off, 15, P.y); mesh_st(off, 16, P.z);
                mesh_st(off, 17, N.x); mesh_st(off, 18, N.y); mesh_st(off, 19, N.z);
            }
            mesh_st(off, 20, __int_as_float(node)); mesh_st(off, 21, __uint_as_float((uint32_t)key)); mesh_st(off, 22, __uint_as_float((uint32_t)(key >> 32)));
            mesh_st(off, 23, __uint_as_float(mpos));
        }
        nmesh += n;
    };

    // radiance -> the path's pixel: of the frame, or of its iteration's accumulator plane (owned rows only)
    auto splat = [&](uint32_t pv, f3 L) {
        const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
        float *base = reinterpret_cast<float *>((uintptr_t)((unsigned long long)park[0] | ((unsigned long long)park[1] << 32)));
        size_t off = (size_t)pixel * 3;
        if (NEE || a.batch > 1u) {
            const uint32_t W = park[4];
            const uint32_t y = (uint32_t)(((unsigned long long)pixel * park[6]) >> park[7]);
            const uint32_t x = pixel - y * W;
            const uint32_t ly = (uint32_t)(((unsigned long long)(y - park[5]) * park[8]) >> park[9]);
            off = (size_t)slot * (size_t)((unsigned long long)park[2] | ((unsigned long long)park[3] << 32)) + (size_t)(ly * W + x) * 3;
        }
        float *px = base + off;
        (void)unsafeAtomicAdd(px, L.x); (void)unsafeAtomicAdd(px + 1, L.y); (void)unsafeAtomicAdd(px + 2, L.z);
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
        if (++turns > pa.turn_limit) { if (lane == 0) *pa.error = 3u; break; }                // never reached; bounds a broken build
        // the scheduler's state is wave-uniform by construction; saying so keeps its arithmetic and its branches on the scalar unit
        nbox = __builtin_amdgcn_readfirstlane(nbox); nsph = __builtin_amdgcn_readfirstlane(nsph); sp = __builtin_amdgcn_readfirstlane(sp);
        jobpos = __builtin_amdgcn_readfirstlane(jobpos); jobend = __builtin_amdgcn_readfirstlane(jobend);
        round = __builtin_amdgcn_readfirstlane(round); ctr = __builtin_amdgcn_readfirstlane(ctr); dry = __builtin_amdgcn_readfirstlane(dry);
        int act;                                           // 0 FRESH from the stack, 3 FRESH camera rays, 1 TEST cubes, 2 TEST spheres, 4 MESH
        if constexpr (MESH) nmesh = __builtin_amdgcn_readfirstlane(nmesh);
        if (MESH && nmesh >= kMeshTurn) act = 4;
        else if (nbox >= 64u) act = 1;
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
                else if (nbox + nsph + nmesh) act = (nmesh > nbox && nmesh > nsph) ? 4 : nbox >= nsph ? 1 : 2;
                else break;
            }
        } else act = nbox >= nsph ? 1 : 2;

        if (act == 0 || act == 3) {
            // ---------------------------------------------------------------- FRESH
            f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1.0f, 1.0f, 1.0f);
            uint32_t pv = 0u, level = NEE ? (1u << 9) : 0u;                                // (NEE: the level WORD; a camera ray counts emission)
            f3 acc = mk(0, 0, 0), nd = mk(0, 0, 0);                                        // NEE only
            float cos_s = 0.0f, invpdf = 0.0f, dist2 = 0.0f;
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
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, PT_SELF_SCOPE);        // the wave's own stack stores have landed (vmcnt 0) ...
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PT_SELF_SCOPE);        // ... before they are read back through the same L1
                if (valid) {
                    o = mk(ring_ld(off, 0), ring_ld(off, 1), ring_ld(off, 2));
                    d = mk(ring_ld(off, 3), ring_ld(off, 4), ring_ld(off, 5));
                    thr = mk(ring_ld(off, 6), ring_ld(off, 7), ring_ld(off, 8));
                    pv = __float_as_uint(ring_ld(off, 9));
                    level = __float_as_uint(ring_ld(off, 10));
                    if constexpr (NEE) {
                        acc = mk(ring_ld(off, 11), ring_ld(off, 12), ring_ld(off, 13));
                        nd = mk(ring_ld(off, 14), ring_ld(off, 15), ring_ld(off, 16));
                        cos_s = ring_ld(off, 17); invpdf = ring_ld(off, 18); dist2 = ring_ld(off, 19);
                    }
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
#ifndef PT_MESH_STATS
            atomicAdd(&g_cull_stats[5], (unsigned long long)__popc(mask));
            if (act == 3) qstat(6, 1ull);
#endif
#endif
            mask &= ~(1u << next_j);
            const bool tobox = push && ((boxbits >> next_j) & 1u);
            const bool tomesh = MESH && push && ((meshbits >> next_j) & 1u);
            if constexpr (MESH) mesh_push(tomesh, o, d, thr, pv, mask, next_j | (level << 8), kInf, -1, -1, mk(0, 0, 0), mk(0, 0, 0), 0, kMeshNoHit, 0u);
            const u64 bb = __ballot(tobox), sb = __ballot(push && !tobox && !tomesh);
            if (bb | sb) {
                if (push && !tomesh) {
                    const uint32_t pos = tobox ? nbox + wave_rank(bb) : kPCap - 1u - (nsph + wave_rank(sb));
                    float *r = q + pos;
                    r[0 * kPCap] = o.x; r[1 * kPCap] = o.y; r[2 * kPCap] = o.z;
                    r[3 * kPCap] = d.x; r[4 * kPCap] = d.y; r[5 * kPCap] = d.z;
                    r[6 * kPCap] = thr.x; r[7 * kPCap] = thr.y; r[8 * kPCap] = thr.z;
                    r[9 * kPCap] = __uint_as_float(pv);
                    r[10 * kPCap] = __uint_as_float(mask);
                    r[11 * kPCap] = __uint_as_float(next_j | (level << 8));
                    if constexpr (NEE) {
                        park_st(pos, PX + 0, acc.x); park_st(pos, PX + 1, acc.y); park_st(pos, PX + 2, acc.z);
                        park_st(pos, PX + 3, nd.x); park_st(pos, PX + 4, nd.y); park_st(pos, PX + 5, nd.z);
                        park_st(pos, PX + 6, cos_s); park_st(pos, PX + 7, invpdf); park_st(pos, PX + 8, dist2);
                    }
                }
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (NEE) {
                // a ray without any candidate: a path ends here with what it has gathered; a shadow ray has simply not reached its
                // emitter, and the scattered ray it held back goes on (next level, emission not counted after a diffuse event)
                const bool gone = valid && !push;
                const bool sh = gone && (level & 0x100u) != 0u;
                if (gone && !sh && (acc.x != 0.0f || acc.y != 0.0f || acc.z != 0.0f)) splat(pv, acc);
                const u64 cb = __ballot(sh);
                if (cb) {
                    if (sh) {
                        const uint32_t off = woff + sp + wave_rank(cb);
                        ring_st(off, 0, o.x); ring_st(off, 1, o.y); ring_st(off, 2, o.z);
                        ring_st(off, 3, nd.x); ring_st(off, 4, nd.y); ring_st(off, 5, nd.z);
                        ring_st(off, 6, thr.x); ring_st(off, 7, thr.y); ring_st(off, 8, thr.z);
                        ring_st(off, 9, __uint_as_float(pv));
                        ring_st(off, 10, __uint_as_float((level & 0xFFu) + 1u));
                        ring_st(off, 11, acc.x); ring_st(off, 12, acc.y); ring_st(off, 13, acc.z);
                    }
                    sp += (uint32_t)__popcll(cb);                     // (the group came off the stack: at most as many go back)
                }
            }
            continue;
        }

        // -------------------------------------------------------------------- TEST (one type per group, any levels) / MESH
        const bool isb = act == 1;
        bool valid;
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(0, 0, 0);
        uint32_t pv = 0u, mask = 0u, level = 0u;
        f3 acc = mk(0, 0, 0), nd = mk(0, 0, 0);                               // NEE only
        float cos_s = 0.0f, invpdf = 0.0f, dist2 = 0.0f;
        int j = 0;
        float best = kInf;
        int hit = -1, face = -1;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 0);
        uint32_t lw = 0u;                                                     // NEE: the level word (level | shadow << 8 | count-emission << 9 | light << 10)
        if (MESH && act == 4) {
            if constexpr (MESH) {
                // ------------------------------------------------------------ MESH: the top 64 rays of the mesh stack
                const uint32_t cnt = nmesh < 64u ? nmesh : 64u;
                valid = lane < cnt;
                nmesh -= cnt;
                const uint32_t off = moff + nmesh + lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, PT_SELF_SCOPE);        // the wave's own stack stores have landed (vmcnt 0) ...
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PT_SELF_SCOPE);        // ... before they are read back through the same L1
                int node = -1;
                unsigned long long key = kMeshNoHit;
                uint32_t mpos = 0u, jl = 0u;
                if (valid) {                                                  // what the traversal needs; the rest of the record after it
                    o = mk(mesh_ld(off, 0), mesh_ld(off, 1), mesh_ld(off, 2));
                    d = mk(mesh_ld(off, 3), mesh_ld(off, 4), mesh_ld(off, 5));
                    jl = __float_as_uint(mesh_ld(off, 11));
                    node = __float_as_int(mesh_ld(off, 20));
                    key = (unsigned long long)__float_as_uint(mesh_ld(off, 21)) | ((unsigned long long)__float_as_uint(mesh_ld(off, 22)) << 32);
                    mpos = __float_as_uint(mesh_ld(off, 23));
                }
                j = (int)(jl & 0xFFu);
                level = jl >> 8;
                const GeomRec *gr = lg + j;
                const unsigned long long mbase = ((unsigned long long)__float_as_uint(gr->bmax[3]) << 32) | (unsigned long long)__float_as_uint(gr->bmin[3]);
                const MeshNode *nodes = reinterpret_cast<const MeshNode *>(mbase);
                const unsigned long long tbase = mbase + (unsigned long long)(uint32_t)gr->inside_hits;
                const f3 ro = mul_point(gr->inv, o);                          // (mesh_test's own first lines)
                const f3 rd = normalize(mul_vector(gr->inv, d));
                const CullRay cr = make_cull_ray(ro, rd);
                const int oct = (rd.x < 0.0f ? 1 : 0) | (rd.y < 0.0f ? 2 : 0) | (rd.z < 0.0f ? 4 : 0);      // near children first (MeshNode)
                mkey[lane] = key;
                mposn[lane] = mpos;
                uint32_t npairs = 0u;
                const bool more = nmesh >= 32u;                               // other rays wait: a thin WALK gives way to them
#ifdef PT_MESH_STATS
                if (lane == 0) { atomicAdd(&g_cull_stats[0], 1ull); atomicAdd(&g_cull_stats[1], (unsigned long long)cnt); }
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (;;) {
                    const u64 wb = __ballot(node >= 0);
                    const uint32_t nw = (uint32_t)__popcll(wb);
                    int op;                                                   // 0 WALK, 1 TRI
                    if (npairs >= 64u) op = 1;
                    else if (nw == 0u || (more && nw < kMeshMinWalk)) { if (npairs) op = 1; else break; }
                    else op = 0;
                    if (op == 0) {
                        // -------------------------------------------------------- WALK: lane = ray, one node of its mesh's threaded BVH
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (uint32_t rep = 0; rep < kMeshWalkSteps; ++rep) {               // (several nodes per trip of the dispatcher)
#ifdef PT_MESH_STATS
                            { const u64 sb2 = __ballot(node >= 0); if (lane == 0) { atomicAdd(&g_cull_stats[2], 1ull); atomicAdd(&g_cull_stats[3], (unsigned long long)__popcll(sb2)); } }
#endif
                            bool leafhit = false;
                            uint32_t e = 0u;
                            if (node >= 0) {
                                const float bt = __uint_as_float((uint32_t)(mkey[lane] >> 32));      // the ray's best triangle so far
                                // (the blob is global memory: saying so gives global_load instead of flat_load, which also counts as an LDS access)
                                typedef float nf4 __attribute__((ext_vector_type(4)));
                                typedef const __attribute__((address_space(1))) nf4 *gf4;
                                typedef const __attribute__((address_space(1))) int *gi1;
                                const gf4 np = (gf4)(uintptr_t)(nodes + node);
                                const nf4 lo = np[0];                         // bmin.xyz, leaf
                                const nf4 hi = np[1];                         // bmax.xyz, far
                                const int skip = ((gi1)(uintptr_t)(nodes + node))[8 + oct];      // where this ray's octant goes on after the node
                                const float bl[3] = {lo.x, lo.y, lo.z}, bh[3] = {hi.x, hi.y, hi.z};
                                float tn;
                                const bool in = cull_box(bl, bh, cr, tn) && !(tn > bt);
                                const int leaf = __float_as_int(lo.w), far = __float_as_int(hi.w);
                                if (!in) node = skip;
                                else if (leaf < 0) node = ((oct >> (far & 3)) & 1) ? (far >> 2) : node + 1;      // the near child first
                                else {
                                    leafhit = true;
                                    e = lane | ((((uint32_t)leaf >> 27) - 1u) << 6) | (((uint32_t)leaf & 0xFFFFFFu) << 8);
                                    node = skip;
                                }
                            }
                            const u64 lb = __ballot(leafhit);
                            if (leafhit) mpair[npairs + wave_rank(lb)] = e;
                            npairs += (uint32_t)__popcll(lb);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    } else {
                        // -------------------------------------------------------- TRI: lane = one (ray, triangle) pair; an entry with triangles
                        // left in its leaf returns for the next one
                        const uint32_t n = npairs < 64u ? npairs : 64u;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const bool tv = lane < n;
                        uint32_t e = 0u;
                        if (tv) e = mpair[npairs - n + lane];
                        npairs -= n;
                        const uint32_t r = e & 63u, rem = (e >> 6) & 3u, tri = e >> 8;
                        const int ra = (int)(r << 2);
                        const f3 pro = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(ro.x))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(ro.y))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(ro.z))));
                        const f3 prd = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(rd.x))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(rd.y))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(ra, __float_as_int(rd.z))));
                        const unsigned long long ptb = (unsigned long long)(uint32_t)__builtin_amdgcn_ds_bpermute(ra, (int)(uint32_t)tbase) |
                                                       ((unsigned long long)(uint32_t)__builtin_amdgcn_ds_bpermute(ra, (int)(uint32_t)(tbase >> 32)) << 32);
#ifdef PT_MESH_STATS
                        if (lane == 0) { atomicAdd(&g_cull_stats[4], (unsigned long long)n); atomicAdd(&g_cull_stats[5], 1ull); }
#endif
                        unsigned long long mykey = kMeshNoHit;
                        bool ok = false;
                        if (tv) {
                            typedef float nf4 __attribute__((ext_vector_type(4)));
                            typedef const __attribute__((address_space(1))) nf4 *gf4;
                            const gf4 tp = (gf4)(uintptr_t)(ptb + (unsigned long long)tri * sizeof(MeshTri));
                            const nf4 va = tp[0], vb = tp[1], vc = tp[2];
                            const float t = triangle_test(mk(va.x, va.y, va.z), mk(vb.x, vb.y, vb.z), mk(vc.x, vc.y, vc.z), pro, prd);
                            ok = t > 0.0f;
                            mykey = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)__float_as_int(va.w);
                        }
                        __builtin_amdgcn_wave_barrier();                      // every entry is read before any is overwritten
                        if (ok) (void)atomicMin(&mkey[r], mykey);             // (t, index in the file): nearest, ties to the earlier triangle
                        if (ok && mkey[r] == mykey) mposn[r] = tri;           // the winner so far says where its triangle sits
                        const bool again = tv && rem != 0u;
                        const u64 gb = __ballot(again);
                        if (again) mpair[npairs + wave_rank(gb)] = r | ((rem - 1u) << 6) | ((tri + 1u) << 8);
                        npairs += (uint32_t)__popcll(gb);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                key = mkey[lane];
                mpos = mposn[lane];
                // the rest of the record: the best hit so far
                uint32_t hf = 0u;
                if (valid) {
                    thr = mk(mesh_ld(off, 6), mesh_ld(off, 7), mesh_ld(off, 8));
                    pv = __float_as_uint(mesh_ld(off, 9));
                    mask = __float_as_uint(mesh_ld(off, 10));
                    best = mesh_ld(off, 12);
                    hf = __float_as_uint(mesh_ld(off, 13));
                }
                hit = (int)(hf & 0xFFu) - 1;
                face = (int)((hf >> 8) & 0xFFu) - 1;
                if (valid && hit >= 0) {
                    P = mk(mesh_ld(off, 14), mesh_ld(off, 15), mesh_ld(off, 16));
                    N = mk(mesh_ld(off, 17), mesh_ld(off, 18), mesh_ld(off, 19));
                }
                // an interrupted traversal goes back with its cursor (every store below follows the load of the same field)
                const bool unf = valid && node >= 0;
#ifdef PT_MESH_STATS
                { const u64 ub2 = __ballot(unf); if (lane == 0) atomicAdd(&g_cull_stats[6], (unsigned long long)__popcll(ub2)); }
#endif
                mesh_push(unf, o, d, thr, pv, mask, jl, best, hit, face, P, N, node, key, mpos);
                if (unf) { hit = -1; mask = 0u; }                             // (its best hit so far went with it: nothing to shade here)
                const bool fin = valid && node < 0;
                if (fin && key != kMeshNoHit) {
                    const float4 ngv = reinterpret_cast<const float4 *>(tbase + (unsigned long long)mpos * sizeof(MeshTri))[3];
                    f3 p, nn;
                    const float depth = mesh_finish(gr->inv, gr->xf, o, ro, rd, __uint_as_float((uint32_t)(key >> 32)), mk(ngv.x, ngv.y, ngv.z), p, nn);
                    if (depth > -PT_EPSILON && (depth < best || (depth == best && j < hit))) { best = depth; hit = j; P = p; N = nn; face = -1; }
                }
                valid = fin;
                lw = level;
            }
        } else {
        const uint32_t have = isb ? nbox : nsph;
        const uint32_t cnt = have < 64u ? have : 64u;
        valid = lane < cnt;
        const uint32_t pos = isb ? (have - cnt + lane) : (kPCap - 1u - (have - cnt + lane));
        if (isb) nbox -= cnt; else nsph -= cnt;
#ifdef PT_CULL_STATS
        qstat(isb ? 10 : 12, 1ull); qstat(isb ? 11 : 13, (unsigned long long)cnt);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (NEE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, PT_SELF_SCOPE);            // the parked words of earlier groups have landed (vmcnt 0) ...
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PT_SELF_SCOPE);            // ... before they are read back through the same L1
        }
        if (valid) {
            const float *r = q + pos;
            o = mk(r[0 * kPCap], r[1 * kPCap], r[2 * kPCap]);
            d = mk(r[3 * kPCap], r[4 * kPCap], r[5 * kPCap]);
            thr = mk(r[6 * kPCap], r[7 * kPCap], r[8 * kPCap]);
            pv = __float_as_uint(r[9 * kPCap]);
            mask = __float_as_uint(r[10 * kPCap]);
            const uint32_t jl = __float_as_uint(r[11 * kPCap]);
            if constexpr (NEE) {
                acc = mk(park_ld(pos, PX + 0), park_ld(pos, PX + 1), park_ld(pos, PX + 2));
                nd = mk(park_ld(pos, PX + 3), park_ld(pos, PX + 4), park_ld(pos, PX + 5));
                cos_s = park_ld(pos, PX + 6); invpdf = park_ld(pos, PX + 7); dist2 = park_ld(pos, PX + 8);
            }
            j = (int)(jl & 0xFFu);
            level = jl >> 8;
        }
        lw = level;
        if constexpr (NEE) level &= 0xFFu;
        __builtin_amdgcn_wave_barrier();
        {
            const GeomRec *gr = lg + j;
            float depth = -1.0f;
            if (isb) { if (valid) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, P, face); }
            else { if (valid) depth = sphere_test(gr->inv, gr->xf, o, d, P, N); }
            const bool wins = valid && depth > -PT_EPSILON && depth < kInf;
            best = wins ? depth : kInf;
            hit = wins ? j : -1;
        }
        }
        bool tomesh = false;                                                  // MESH: the next candidate is a mesh: the ray goes on the mesh stack
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
                if constexpr (MESH) {
                    if (active && ((meshbits >> j) & 1u)) { tomesh = true; active = false; }
                    if (!__any(active)) break;
                }
                const bool jb = (boxbits >> j) & 1u;
                const GeomRec *gr = lg + j;
                f3 p = mk(0, 0, 0), nn = mk(0, 0, 0);
                int fc = -1;
                float depth = -1.0f;
                if (__any(active && jb)) { if (active && jb) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, p, fc); }
                if (__any(active && !jb)) { if (active && !jb) depth = sphere_test(gr->inv, gr->xf, o, d, p, nn); }
                const bool wins = active && depth > -PT_EPSILON && (depth < best || (depth == best && j < hit));
                if (wins) { best = depth; hit = j; P = p; N = nn; face = fc; }
            }
        }
        if constexpr (MESH) {
            mesh_push(tomesh, o, d, thr, pv, mask, (uint32_t)j | (level << 8), best, hit, face, P, N, 0, kMeshNoHit, 0u);
            if (tomesh) hit = -1;                                             // (not shaded in this turn)
        }
#ifdef PT_CULL_STATS
        qstat(14, (unsigned long long)__popcll(__ballot(hit >= 0)));
        qstat(7, (unsigned long long)__popcll(__ballot(hit >= 0 && level + 1u >= D)));
#endif
        if constexpr (NEE) {
            // ---------------------------------------------------------------- direct light: shadow rays resolve, paths carry their radiance
            const bool sh = valid && (lw & 0x100u) != 0u;
            bool to_stack = false;                                            // this lane's path goes on: (o, d, thr, next level word) below
            uint32_t next_lw = 0u;
            if (sh) {
                // the shadow ray's nearest hit decides (oracle direct_light): the sampled emitter itself, not hidden by a nearer face of it
                const uint32_t lid = (lw >> 10) & 0xFFu;
                const float dist = __builtin_sqrtf(dist2);
                const float tol = 1e-3f * (dist > 1.0f ? dist : 1.0f);
                if (hit == (int)lid && (best + tol >= dist)) {
                    const f3 Nh = ((boxbits >> lid) & 1u) ? box_face_normal(lg[lid].xf, face) : N;
                    const float nl2 = dot(Nh, Nh);
                    if (nl2 > 0.0f) {
                        const float cos_l = fabsf(dot(Nh, d)) / __builtin_sqrtf(nl2);
                        const float geomf = (((cos_s * cos_l) * invpdf) / (PT_PI * dist2)) * (float)a.nlights;
                        const MatRec ml = lm[lg[lid].mat];
                        const f3 Le = mk(ml.color[0], ml.color[1], ml.color[2]) * ml.emittance;
                        const f3 Cc = (thr * Le) * geomf;
                        acc = mk(acc.x + Cc.x, acc.y + Cc.y, acc.z + Cc.z);
                    }
                }
                d = nd;                                                       // the scattered ray it held back goes on from the same origin
                next_lw = level + 1u;                                         // (after a diffuse event emission is not counted)
                to_stack = true;
            }
            bool alive = false, ended = valid && !sh && hit < 0;
            if (valid && !sh && hit >= 0) {
                const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
                const MatRec m = lm[lg[hit].mat];
                if (level + 1u >= D && !(m.emittance > 0.0f)) {
                    alive = true; ended = true;                               // depth exhausted: alive, contributes 0
                } else {
                    const uint32_t iteration = a.iteration + slot;
                    uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + level));
                    st = lcg_next(st); const float u_sel = u01(st);
                    st = lcg_next(st); const float xi1 = u01(st);
                    st = lcg_next(st); const float xi2 = u01(st);
                    const f3 d_in = d;
                    f3 L = mk(0.0f, 0.0f, 0.0f);
                    int code = 4;
                    const bool hb = (boxbits >> hit) & 1u;
                    const f3 Ng = hb ? box_face_normal(lg[hit].xf, face) : N;     // the geometric normal the light sample is oriented by
                    if (__any(hb)) { if (hb) code = scatter_box(m, P, face, lf + 3 * hit, u_sel, xi1, xi2, o, d, thr, L); }
                    if (__any(!hb)) { if (!hb) code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, L); }
                    if (code == 3) {
                        if (lw & 0x200u) acc = mk(acc.x + L.x, acc.y + L.y, acc.z + L.z);
                        emitted++;
                    }
                    alive = code <= 2;
                    ended = !alive;
                    if (alive) {
                        to_stack = true;
                        next_lw = (level + 1u) | ((code == 1 || code == 2) ? 0x200u : 0u);
                        if (code == 0 && a.nlights > 0u) {
                            // one shadow ray to a point on a random emitter (the reference's getRandomPointOnCube / Sphere)
                            st = lcg_next(st); const float u_l = u01(st);
                            st = lcg_next(st); const float seedf = (float)(st & 0xFFFFFFu);
                            int li = (int)(u_l * (float)a.nlights);
                            if (li > (int)a.nlights - 1) li = (int)a.nlights - 1;
                            const int lid = (int)a.lights[li];
                            f3 Q;
                            float ipdf;
                            const bool ok = sample_light(lg[lid].xf, lg[lid].type, seedf, Q, ipdf);
                            const f3 wv = Q - o;
                            const float d2 = dot(wv, wv);
                            if (ok && d2 > 0.0f) {
                                const float dist = __builtin_sqrtf(d2);
                                const f3 w = wv * (1.0f / dist);
                                const f3 n = Ng * (1.0f / __builtin_sqrtf(dot(Ng, Ng)));
                                const f3 nf = (dot(n, d_in) > 0.0f) ? neg(n) : n;
                                const float cs = dot(nf, w);
                                if (cs > 0.0f) {
                                    nd = d; d = w;                            // the shadow ray first; the scattered direction waits in the record
                                    cos_s = cs; invpdf = ipdf; dist2 = d2;
                                    next_lw = level | 0x100u | ((uint32_t)lid << 10);
                                }
                            }
                        }
                    }
                }
            }
            if (alive) atomicAdd(&lsurv[level + 1u], 1u);
            // a path that ends here leaves what it has gathered in its pixel of its iteration's plane
            if (ended && (acc.x != 0.0f || acc.y != 0.0f || acc.z != 0.0f)) splat(pv, acc);
            const u64 ballot = __ballot(to_stack);
            if (ballot) {
                const uint32_t n = (uint32_t)__popcll(ballot);
                if (sp + n > KS) { if (lane == 0) *pa.error = 2u; }
                else {
                    if (to_stack) {
                        const uint32_t off = woff + sp + wave_rank(ballot);
                        ring_st(off, 0, o.x); ring_st(off, 1, o.y); ring_st(off, 2, o.z);
                        ring_st(off, 3, d.x); ring_st(off, 4, d.y); ring_st(off, 5, d.z);
                        ring_st(off, 6, thr.x); ring_st(off, 7, thr.y); ring_st(off, 8, thr.z);
                        ring_st(off, 9, __uint_as_float(pv));
                        ring_st(off, 10, __uint_as_float(next_lw));
                        ring_st(off, 11, acc.x); ring_st(off, 12, acc.y); ring_st(off, 13, acc.z);
                        ring_st(off, 14, nd.x); ring_st(off, 15, nd.y); ring_st(off, 16, nd.z);
                        ring_st(off, 17, cos_s); ring_st(off, 18, invpdf); ring_st(off, 19, dist2);
                    }
                    sp += n;
                }
            }
            continue;
        }
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
        bool onward = alive && level + 1u < D;
        if (pa.tap_level != 0u) {                                             // parity hook: the rays entering bounce tap_level leave here
            const bool tapped = onward && level + 1u == pa.tap_level;
            const u64 tb = __ballot(tapped);
            if (tb) {
                uint32_t base = 0u;
                if (lane == 0) base = atomicAdd(pa.tap_count, (uint32_t)__popcll(tb));
                base = __builtin_amdgcn_readfirstlane(base);
                if (tapped) {
                    float *t = pa.tap + base + wave_rank(tb);
                    const size_t cap = pa.tap_cap;
                    t[0 * cap] = o.x; t[1 * cap] = o.y; t[2 * cap] = o.z; t[3 * cap] = d.x; t[4 * cap] = d.y; t[5 * cap] = d.z;
                    t[6 * cap] = thr.x; t[7 * cap] = thr.y; t[8 * cap] = thr.z; t[9 * cap] = __uint_as_float(pv);
                }
            }
            onward = onward && !tapped;
        }
        const u64 ballot = __ballot(onward);
        if (ballot) {
            const uint32_t n = (uint32_t)__popcll(ballot);
            if (sp + n > KS) { if (lane == 0) *pa.error = 2u; }        // never: see the bound above
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

// ------------------------------------------------------------------ host side ---------
template <int CAP>
static const void *path_fn(bool mesh, bool nee) {
    return nee ? reinterpret_cast<const void *>(&k_path_q<false, true, CAP>)
               : mesh ? reinterpret_cast<const void *>(&k_path_q<true, false, CAP>) : reinterpret_cast<const void *>(&k_path_q<false, false, CAP>);
}
template <int CAP>
static void path_go(bool mesh, bool nee, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const PathArgs &pa,
                    const GeomRec *g, const MatRec *m, const QTables &qt) {
    if (nee) hipLaunchKernelGGL((k_path_q<false, true, CAP>), dim3(grid), dim3(kBlock), lds, st, a, pa, g, m, qt);
    else if (mesh) hipLaunchKernelGGL((k_path_q<true, false, CAP>), dim3(grid), dim3(kBlock), lds, st, a, pa, g, m, qt);
    else hipLaunchKernelGGL((k_path_q<false, false, CAP>), dim3(grid), dim3(kBlock), lds, st, a, pa, g, m, qt);
}

hipError_t path_setup(bool mesh, bool nee, int cap, uint32_t lds_bytes, int *blocks_per_cu) {
    const void *fn = nullptr;
#define PT_X(C) if (cap == C) fn = path_fn<C>(mesh, nee);
    PT_PATH_CAPS(PT_X)
#undef PT_X
    if (!fn) return hipErrorInvalidValue;
    if (lds_bytes > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kBlock, lds_bytes) != hipSuccess || occ < 1) occ = 2;
    *blocks_per_cu = occ;
    return hipSuccess;
}

void path_launch(bool mesh, bool nee, int cap, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const PathArgs &pa,
                 const GeomRec *g, const MatRec *m, const QTables &qt) {
    switch (cap) {
#define PT_X(C) case C: path_go<C>(mesh, nee, grid, lds, st, a, pa, g, m, qt); break;
    PT_PATH_CAPS(PT_X)
#undef PT_X
    default: break;                                        // (path_setup has refused any other capacity)
    }
}

#ifdef PT_CULL_STATS
void cull_stats_path(unsigned long long *acc16) {
    unsigned long long v[16];
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_cull_stats), sizeof v) == hipSuccess) for (int i = 0; i < 16; ++i) acc16[i] += v[i];
}
#endif

}  // namespace ptk
