// pt_k_queue.hip -- `ordering = 1`: the typed work queues, one launch per bounce (<= 32 primitives).  The description
// of the two stages is in pt_kernels.hpp next to the queue constants; k_path_q (pt_k_path.hip) runs the same stages
// over whole paths.
#include "pt_kernels.hpp"

namespace ptk {

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

// ------------------------------------------------------------------ host side ---------
hipError_t queue_setup(bool mesh, uint32_t lds_bytes, int *blocks_per_cu) {
    const void *fns[8] = {
        reinterpret_cast<const void *>(&k_bounce_q<false, false>), reinterpret_cast<const void *>(&k_bounce_q<true, false>),
        reinterpret_cast<const void *>(&k_bounce_q<false, true>), reinterpret_cast<const void *>(&k_bounce_q<true, true>),
        reinterpret_cast<const void *>(&k_bounce_q<false, false, true>), reinterpret_cast<const void *>(&k_bounce_q<true, false, true>),
        reinterpret_cast<const void *>(&k_bounce_q<false, true, true>), reinterpret_cast<const void *>(&k_bounce_q<true, true, true>)};
    if (lds_bytes > 64u * 1024u)
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[(mesh ? 4 : 0) + i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fns[mesh ? 4 : 0], kBlock, lds_bytes) != hipSuccess || occ < 1) occ = 2;
    *blocks_per_cu = occ;
    return hipSuccess;
}

namespace {
template <bool MESH>
void queue_launch_m(bool last, bool gen, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const GeomRec *g, const MatRec *m, const QTables &qt) {
    if (gen) {
        if (last) hipLaunchKernelGGL((k_bounce_q<true, true, MESH>), dim3(grid), dim3(kBlock), lds, st, a, g, m, qt);
        else hipLaunchKernelGGL((k_bounce_q<false, true, MESH>), dim3(grid), dim3(kBlock), lds, st, a, g, m, qt);
    } else {
        if (last) hipLaunchKernelGGL((k_bounce_q<true, false, MESH>), dim3(grid), dim3(kBlock), lds, st, a, g, m, qt);
        else hipLaunchKernelGGL((k_bounce_q<false, false, MESH>), dim3(grid), dim3(kBlock), lds, st, a, g, m, qt);
    }
}
}  // namespace

void queue_launch(bool mesh, bool last, bool gen, int grid, uint32_t lds, hipStream_t st, const SegArgs &a,
                  const GeomRec *g, const MatRec *m, const QTables &qt) {
    if (mesh) queue_launch_m<true>(last, gen, grid, lds, st, a, g, m, qt);
    else queue_launch_m<false>(last, gen, grid, lds, st, a, g, m, qt);
}

#ifdef PT_CULL_STATS
void cull_stats_queue(unsigned long long *acc16) {
    unsigned long long v[16];
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_cull_stats), sizeof v) == hipSuccess) for (int i = 0; i < 16; ++i) acc16[i] += v[i];
}
#endif

}  // namespace ptk
