// pt_k_wide.hip -- `ordering = 2` for scenes of MORE THAN 32 analytic primitives, any number: whole paths in ONE launch per
// group of iterations, with every expensive step on dense, type-pure waves.
//
// The nearest-hit loop it replaces is type- and count-agnostic (/root/reference/src/raytraceKernel.cu:134-153: every
// primitive, first strictly nearer wins, ties to the lower index; any numberOfGeoms, :192-194).  The lock-step kernel for
// these scenes (k_bounce_seg, pt_k_seg.hip) finds the same hits with wave-uniform passes over the bounds -- all of them beyond
// 256 primitives -- and per-lane loops that leave the wave mostly idle.  Here a ray only meets the primitives near its own
// path, and the unit of work is no longer "a ray":
//
//   FRESH   64 rays (the top of the wave's stack of survivors, or a job of camera rays) take a RAY SLOT each in the
//           wave's LDS region (origin + direction, an 8-entry candidate list, best depth / hit so far) and park
//           throughput, pixel word and level in the slot's payload record in global memory.  The few BIG primitives
//           (walls: listed in too many cells to be worth a grid entry) have their bounds tested by every ray in a
//           wave-uniform loop.
//           CAMERA GROUPS (up to 256 primitives): the 64 rays share their origin and span a degree or two, so instead of 64
//           walks the wave tests every primitive's bound once against the group's CONE and the PLANE of its fan (lane = one
//           primitive; pt_kernels.hpp, fan_*), and only the handful it meets have their bounds tested by the rays.
//   WALK    every ray steps through the cells of a uniform grid over the small primitives (3D-DDA, one cell per
//           trip; boundaries are rebuilt from the integer cell index, never accumulated).  A step into a non-empty
//           cell leaves a (ray, cell, axes stepped) entry in a small LDS buffer.
//   CELLS   64 such entries at a time, lane = one entry = one REFERENCE of the cell's list: a primitive that is NEW to
//           the ray there -- decided by integer flags stored with the reference (GridArgs, pt_kernels.hpp), no
//           arithmetic on distances -- goes on as a (ray, primitive) entry, cubes and spheres to the two ends of a
//           second buffer; unless the reference is its cell's last, the entry returns to the buffer for the next
//           one (so a crowded cell never holds a wave in a loop).
//   BOUNDS  64 (ray, primitive) entries of one type, lane = one entry: gather the ray from its slot, test the
//           primitive's own conservative bound, append a candidate to the ray's list as a key (quantised
//           conservative entry distance, id).
//   SELECT  every ray picks its nearest candidate (smallest key) and waits on one of two wave-private stacks of slot
//           ids by that candidate's TYPE.  A ray whose list overflowed (> 8 candidates: rare), or that cannot be
//           walked (non-finite, or so far from the grid that its float error exceeds the margin the cells were filled
//           with), takes the reference loop itself on the spot and waits for one confirming test of the winner.
//   TEST    pops 64 slots of one type: the exact reference test of the current candidate on all lanes; the winner so
//           far is kept as (depth, id, face) in the slot.  Then the next candidate that could still win or tie -- key
//           distance not beyond the best hit -- is selected by a branch-free scan of the 8 keys (no re-evaluation of
//           bounds), and the ray goes back on the stack of THAT candidate's type (hit point and normal of a winning
//           test go to the payload record meanwhile).  A finished ray frees its slot; with a hit its payload record waits
//           for the shading of that TYPE of primitive.
//   SHADE   pops 64 payload records of one hit type: RNG stream of the ray's own level, scatter, emitters -> memory-side
//           float atomics; a survivor goes on one of three stacks in global memory by the length of the walk ahead of it.
//
// Two id formats, one source (template parameter BIG):
//   narrow (33..256 primitives)  byte ids, 16-bit keys and references; the geometry table (37 KB at 256) and the grid are staged
//                                in LDS beside the waves' regions;
//   wide   (more than 256)       ids of up to 21 bits, 32-bit keys (10-bit distance, type, id) and references; the geometry is
//                                gathered from the device array through the vector cache, the grid is staged in LDS when it fits
//                                beside the waves' regions (about 1 100 primitives) and read from global memory otherwise.
//
// WALK, CELLS and BOUNDS interleave under one wave-uniform dispatcher: a buffer is drained whenever it holds 64
// entries, so every stage but the walk itself runs on full waves whatever a single ray meets.  Nothing leaves the
// wave: no barrier, no inter-wave traffic, no pools.  One 1024-thread block per CU shares the tables and the grid.
// Results are the reference loop's: the cells are filled with margins far above the walk's float error, so every primitive
// whose bound the ray enters is found, and every candidate entered before the best hit is tested exactly; image, live counts,
// emitter hits and ray records equal every other kernel's bit for bit (tests/test_gpu_parity.py,
// tests/test_gpu_whole_paths_and_bench.py, tests/test_gpu_many_primitives.py; the index itself: tests/test_grid_cpu.py).
#include "pt_kernels.hpp"

namespace ptk {

#ifdef PT_CULL_STATS
static __device__ unsigned long long g_phase_cycles[16];
static __device__ unsigned long long g_wstats[32];
// [0] FRESH groups [1] valid lanes [2] walk trips [3] walking lanes [4] (ray, cell) entries [5] CELLS chunks [6] lanes [7] -
// [8] - [9] (ray, primitive) entries [10] BOUNDS cube chunks [11] lanes [12] BOUNDS sphere chunks [13] lanes [14] candidates
// [15] overflowed / unwalked rays [16] TEST cube groups [17] lanes [18] TEST sphere groups [19] lanes [20] shaded [21] done
// [22] requeued [23] requeued after a win [24] unwalked rays [25] bound tests of big primitives (lanes) [26] SHADE cube-hit groups [27] lanes
// [28] SHADE sphere-hit groups [29] lanes
__device__ __forceinline__ void wstat(int i, unsigned long long v) { if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g_wstats[i], v); }
#define PT_WSTAT(i, v) wstat((i), (unsigned long long)(v))
#else
#define PT_WSTAT(i, v) do { } while (0)
#endif

#ifndef PT_WALK_FLAT
#define PT_WALK_FLAT 1                  // the walk reads and steps on every lane, walking or not (two exec-masked regions less per trip: 0.8907 -> 0.8847 ms/step)
#endif
#ifndef PT_FAN
#define PT_FAN 1                        // camera groups take one cone test per primitive instead of 64 grid walks (A/B switch; results identical)
#endif

namespace {

// slot fields (SoA, stride R dwords, wave-private LDS): origin, direction | the candidate list | meta word | best depth [| best hit]
// narrow (<= 256 primitives): 8 keys of 16 bits in 4 words; wide (BIG): 8 keys of 32 bits, and the best hit's id in a word of its own
enum : uint32_t { F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_L0 };
template <bool BIG> struct WFmt;
template <> struct WFmt<false> {
    static constexpr uint32_t kListWords = 4, kMeta = F_L0 + 4, kBest = kMeta + 1, kHit = kBest, kFields = 12;   // (kHit unused)
    static constexpr uint32_t kQMax = 254u, kEmpty = 0xFFFFu;           // quantised entry distance 0..254 (255 would collide with the empty key)
    static constexpr uint32_t kSidBits = 8, kRefShift = 8, kRefMask = 0x1FFFu, kEmaskShift = 21;        // (ray, cell reference) entry: sid | ref index | entry flags
};
template <> struct WFmt<true> {
    static constexpr uint32_t kListWords = 8, kMeta = F_L0 + 8, kBest = kMeta + 1, kHit = kBest + 1, kFields = 17;
    static constexpr uint32_t kQMax = 1022u, kEmpty = 0xFFFFFFFFu;      // key = q << 22 | sphere << 21 | id (21 bits)
    static constexpr uint32_t kSidBits = 7, kRefShift = 7, kRefMask = 0x3FFFFu, kEmaskShift = 25;
};
constexpr uint32_t kBigIdMask = 0x1FFFFFu, kBigSphere = 1u << 21;
// payload fields (SoA, stride R floats, global memory, per wave)
// payload record (global memory, per wave): ONE 64-byte line per ray, four 16-byte quarters -- a lane of a TEST or SHADE group
// holds some ray of the wave, so field-major arrays would cost it a different cache line per field:
//   Q_THR throughput.xyz, pixel word | Q_DIR level, direction.xyz | Q_HIT best hit's point.xyz, hit | (face + 1) << 8 (wide ids: << 24) | Q_NRM its normal.xyz (spheres), -
enum : uint32_t { Q_THR = 0, Q_DIR = 1, Q_HIT = 2, Q_NRM = 3 };
static_assert(kWPayload == 16, "payload record: one 64-byte line");
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static_assert(kWalkBins == 3, "the survivors' stacks are written out as sp0, sp1, sp2");
static_assert(100 * 4 + sizeof(GridArgs) <= kCtrlBytes, "the parked GridArgs must fit behind the survivors' counters in the control block");

constexpr uint32_t kMetaHasHit = 1u << 27;      // narrow slot word while a ray waits for a TEST: candidate | best hit << 8 | (face + 1) << 16 | payload id << 19 | this bit
                                                // wide: candidate (21 bits) | (face + 1) << 21 | payload id << 24; hit word: best hit | hit is a cube << 30 | has a hit << 31
constexpr uint32_t kListCap = 8;
constexpr uint32_t kBufA = 128;        // (ray, cell) entries: a walk trip adds <= 64 to <= 63 waiting
constexpr uint32_t kBufC = 192;        // (ray, primitive) entries, cubes from the bottom, spheres from the top: a CELLS trip adds <= 64 to <= 63 + 63

// quantised conservative entry distance: floor(max(tn, 0) * qscale), 0..kQMax
template <bool BIG>
__device__ __forceinline__ uint32_t quant_tn(float tn, float qscale) {
    const float v = fminf(fmaxf(tn, 0.0f) * qscale, (float)WFmt<BIG>::kQMax);
    return (uint32_t)v;
}

// the smallest viable key among the 8 entries of a list (narrow: key = q << 8 | id, 16 bits, two per word; wide: q << 22 | sphere << 21 | id);
// viable: q <= qmax.  Returns the key (kEmpty: none) and its position.
template <bool BIG>
__device__ __forceinline__ uint32_t select_next(const uint32_t *L, uint32_t qmax, uint32_t &pos) {
    uint32_t bestk = WFmt<BIG>::kEmpty, bestp = 0u;
    const uint32_t limit = BIG ? ((qmax << 22) | 0x3FFFFFu) : ((qmax << 8) | 0xFFu);           // keys above it are beyond the best hit (and kEmpty > limit: qmax <= kQMax)
#pragma unroll
    for (uint32_t k = 0; k < kListCap; ++k) {
        const uint32_t key = BIG ? L[k] : ((k & 1u) ? (L[k >> 1] >> 16) : (L[k >> 1] & 0xFFFFu));
        const bool better = key <= limit && key < bestk;
        bestk = better ? key : bestk;
        bestp = better ? k : bestp;
    }
    pos = bestp;
    return bestk;
}

}  // namespace

template <int WAVES, int R, int NP, bool BIG>
__global__ __launch_bounds__(WAVES * 64) void k_path_w(SegArgs a, PathArgs pa, GridArgs ga, const GeomRec *__restrict__ geoms,
                                                        const MatRec *__restrict__ mats, const FaceFrame *__restrict__ frames) {
    static_assert(R % 4 == 0 && NP % 4 == 0 && R <= 256 && NP <= 256 && NP >= R + 64, "slot and payload ids are bytes");
    static_assert(!BIG || R <= 128, "wide pair entries keep the slot id in 7 bits");
    typedef WFmt<BIG> Fm;
    constexpr uint32_t F_META = Fm::kMeta, F_BEST = Fm::kBest, F_HIT = Fm::kHit, NF = Fm::kFields, LW = Fm::kListWords;
    // survivors wait for their next bounce on kWalkBins stacks, sorted by the length of the walk ahead of them, so that the rays of
    // a FRESH group walk about equally far.  New (camera) rays only enter while no stack holds a full wave and 64 payload records
    // are free, so the wave never holds more than NP + 63 * kWalkBins rays in all: no stack outgrows STK
    constexpr uint32_t STK = (uint32_t)((NP + 64 * kWalkBins + 63) / 64 * 64);
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
        *reinterpret_cast<GridArgs *>(ctrl + 100) = ga;    // the walk's set-up reads the grid's figures from here: the scalar file is full,
    }                                                      // and a spilled scalar costs a lane read per use
    // the grid blob goes behind the tables (stage_tables ends with the block's barrier).  Narrow: always; wide: when the host found
    // room for it (ga.in_lds) -- else cells, references and the big list are read from the device copy
    const uint32_t tb = tables_bytes(a.G, a.M, !BIG);
    const bool grid_lds = !BIG || ga.in_lds != 0u;
    const uint32_t grid_lds_bytes = grid_lds ? ga.blob_bytes : 0u;
    if (grid_lds) {
        uint32_t *gdst = reinterpret_cast<uint32_t *>(smem + tb);
        const uint32_t *gsrc = reinterpret_cast<const uint32_t *>(ga.blob);
        for (uint32_t i = threadIdx.x; i < ga.blob_bytes / 4u; i += blockDim.x) gdst[i] = gsrc[i];
    }
    GeomRec *lg_lds;
    MatRec *lm;
    stage_tables(smem, geoms, a.G, mats, a.M, !BIG, lg_lds, lm);
    // the geometry table: LDS (narrow) or the device array itself (wide: per-lane gathers through the vector cache)
    const GeomRec *lg;
    if constexpr (BIG) lg = geoms; else lg = lg_lds;
    const uint32_t *cells = reinterpret_cast<const uint32_t *>(smem + tb);
    const unsigned short *refs = reinterpret_cast<const unsigned short *>(cells + ga.ncells);                 // narrow
    const unsigned char *bigs = reinterpret_cast<const unsigned char *>(refs + ((ga.nrefs + 1u) & ~1u));
    const uint32_t *gcells = reinterpret_cast<const uint32_t *>(ga.blob);                                    // wide, grid in global memory
    const uint32_t *refs32 = cells + ga.ncells, *grefs32 = gcells + ga.ncells;                                // wide
    const int nbig = (int)ga.nbig;
    auto ld_cell = [&](uint32_t i) -> uint32_t { if constexpr (BIG) { return grid_lds ? cells[i] : gcells[i]; } else { return cells[i]; } };
    auto ld_ref32 = [&](uint32_t i) -> uint32_t { return grid_lds ? refs32[i] : grefs32[i]; };
    auto ld_big = [&](int k) -> uint32_t {                                     // wave-uniform index
        if constexpr (BIG) { return grid_lds ? refs32[ga.nrefs + (uint32_t)k] : grefs32[ga.nrefs + (uint32_t)k]; } else { return (uint32_t)bigs[k]; }
    };

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t wslot = blockIdx.x * WAVES + wave;
    const uint32_t nwaves = gridDim.x * WAVES;
    const uint32_t D = pa.depth;
    // the wave's LDS region: NF slot fields x R | byte lists: typed stacks of slot ids waiting for a TEST (cubes up, spheres down),
    // free slots, free payload records, typed stacks of payload ids waiting for SHADE (cube hits up, sphere hits down) | the two pair buffers
    constexpr uint32_t kListDwords = (2u * R + 2u * NP) / 4u;
    uint32_t *wl = reinterpret_cast<uint32_t *>(smem + tb + grid_lds_bytes) + (size_t)wave * (NF * R + kListDwords + kBufA + kBufC);
    float *wf = reinterpret_cast<float *>(wl);
    unsigned char *xstack = reinterpret_cast<unsigned char *>(wl + NF * R), *freel = xstack + R, *pfree = freel + R, *shstack = pfree + NP;
    uint32_t *bufA = wl + NF * R + kListDwords, *bufC = bufA + kBufA;
    for (uint32_t i = lane; i < (uint32_t)R; i += 64u) freel[i] = (unsigned char)i;
    for (uint32_t i = lane; i < (uint32_t)NP; i += 64u) pfree[i] = (unsigned char)i;
    uint32_t nfree = R, npfree = NP;

    // the wave's arena in global memory: the survivors' stacks (kWalkBins x kSFields x STK) and the payload records (kWPayload x NP)
    const uint32_t wave_floats = kWalkBins * kSFields * STK + kWPayload * (uint32_t)NP;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.arena, 0, pa.arena_bytes, 0x00020000);
    const uint32_t woff = wslot * wave_floats;             // in floats; the arena is below 4 GiB (checked by the host)
    const uint32_t poff = woff + kWalkBins * kSFields * STK;
    auto ring_ld = [&](uint32_t off, uint32_t f) -> float { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, f * STK * 4u, 0)); };
    auto ring_st = [&](uint32_t off, uint32_t f, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off * 4u, f * STK * 4u, 0); };
    auto pay_ld = [&](uint32_t pid, uint32_t q) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rs, (poff + pid * kWPayload) * 4u, q * 16u, 0); };
    auto pay_st = [&](uint32_t pid, uint32_t q, float a, float b, float c, float d) {
        u32x4 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); v.z = __float_as_uint(c); v.w = __float_as_uint(d);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (poff + pid * kWPayload) * 4u, q * 16u, 0);
    };

    // entry `pos` (< kListCap) of slot `sid`'s candidate list: quantised entry distance, primitive, its type
    auto list_put = [&](uint32_t sid_, uint32_t pos, uint32_t q, uint32_t prim, bool sphere) {
        if constexpr (BIG) wl[(F_L0 + pos) * R + sid_] = (q << 22) | (sphere ? kBigSphere : 0u) | prim;
        else {
            unsigned short *l16 = reinterpret_cast<unsigned short *>(&wl[(F_L0 + (pos >> 1)) * R + sid_]) + (pos & 1u);
            *l16 = (unsigned short)((q << 8) | prim);
        }
    };
    // entry `pos` leaves the list (L = the list's words as read)
    auto list_clear = [&](uint32_t sid_, uint32_t pos, const uint32_t *L) {
        if constexpr (BIG) wl[(F_L0 + pos) * R + sid_] = 0xFFFFFFFFu;
        else {
            const uint32_t clr = (pos & 1u) ? 0xFFFF0000u : 0x0000FFFFu;
            wl[(F_L0 + (pos >> 1)) * R + sid_] = L[pos >> 1] | clr;
        }
    };
    uint32_t *bank = a.bank ? a.sync->counts_b : a.sync->counts;
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        uint32_t *other = a.bank ? a.sync->counts : a.sync->counts_b;
        a.sync->totals[threadIdx.x] += other[threadIdx.x];
        other[threadIdx.x] = 0u;
        if (threadIdx.x == 0) bank[0] = a.n_rays;
    }

    uint32_t emitted = 0u;
    uint32_t nbox = 0u, nsph = 0u;                         // slots waiting on the two typed stacks
    uint32_t nshb = 0u, nshs = 0u;                         // payload records waiting to be shaded: cube hits, sphere hits
    uint32_t sp0 = 0u, sp1 = 0u, sp2 = 0u;                 // rays on the wave's stacks of survivors: short, middle and long walks ahead
    uint32_t jobpos = 0u, jobend = 0u;
    bool tickets_left = true;
    uint32_t next_ticket = 0u, round = 0u;
    uint32_t ctr = wslot % kTicketCtrs, dry = 0u;
    if (pa.static_rounds == 0u && lane == 0) next_ticket = atomicAdd(pa.ticket + ctr * kTicketStride, 1u);
    uint32_t turns = 0u;
    const float kInf = 100000000000000000.0f;
#ifdef PT_CULL_STATS
    // phase clock of the analysis build: cycles between phase marks, summed per wave (0 schedule, 1 fresh load + big primitives,
    // 2 walk, 3 cells, 4 bounds, 5 select, 6 test load + exact test, 7 next candidate + requeue, 8 shading, 9 survivors)
    // (pinned: the scheduler may not move instructions across a mark, and the mark waits for the phase's LDS / scalar traffic;
    // vector-memory latency shows where the value is used)
    unsigned long long ph_acc[10] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    auto ph_now = [&]() -> unsigned long long {
        unsigned long long t_;
        __builtin_amdgcn_sched_barrier(0);
        __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t_;
    };
    unsigned long long ph_t0 = ph_now();
    int ph_cur = 0;
#define PT_PHASE(i) do { const unsigned long long t1_ = ph_now(); ph_acc[ph_cur] += t1_ - ph_t0; ph_t0 = t1_; ph_cur = (i); } while (0)
#else
#define PT_PHASE(i) PT_MARK("phase" #i)
#endif
    const float qscale = pa.qscale, slack_max = pa.slack_max;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (;;) {
        PT_PHASE(0);
        // the scheduler's state is wave-uniform by construction; saying so keeps its arithmetic on the scalar unit
        nbox = __builtin_amdgcn_readfirstlane(nbox); nsph = __builtin_amdgcn_readfirstlane(nsph); nfree = __builtin_amdgcn_readfirstlane(nfree);
        nshb = __builtin_amdgcn_readfirstlane(nshb); nshs = __builtin_amdgcn_readfirstlane(nshs); npfree = __builtin_amdgcn_readfirstlane(npfree);
        sp0 = __builtin_amdgcn_readfirstlane(sp0); sp1 = __builtin_amdgcn_readfirstlane(sp1); sp2 = __builtin_amdgcn_readfirstlane(sp2);
        jobpos = __builtin_amdgcn_readfirstlane(jobpos); jobend = __builtin_amdgcn_readfirstlane(jobend);
        round = __builtin_amdgcn_readfirstlane(round); ctr = __builtin_amdgcn_readfirstlane(ctr); dry = __builtin_amdgcn_readfirstlane(dry);
        turns = __builtin_amdgcn_readfirstlane(turns);
        if (++turns > pa.turn_limit) { if (lane == 0) *pa.error = 3u; break; }             // never reached; bounds a broken build
        int act;                                           // 0 FRESH from the stack, 3 FRESH camera rays, 1 TEST cubes, 2 TEST spheres, 4 SHADE cube hits, 5 SHADE sphere hits
        // whatever holds a full wave runs; else a fresh group while slots and payload records are free; else the fullest stage
        auto fullest = [&]() -> int {
            uint32_t m = nbox; int w = 1;
            if (nsph > m) { m = nsph; w = 2; }
            if (nshb > m) { m = nshb; w = 4; }
            if (nshs > m) { m = nshs; w = 5; }
            return m ? w : -1;
        };
        if (nshb >= 64u) act = 4;
        else if (nshs >= 64u) act = 5;
        else if (nbox >= 64u) act = 1;
        else if (nsph >= 64u) act = 2;
        else if (nfree >= 64u && npfree >= 64u) {
            if (sp0 >= 64u || sp1 >= 64u || sp2 >= 64u) act = 0;
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
                else if (sp0 | sp1 | sp2) act = 0;
                else { act = fullest(); if (act < 0) break; }
            }
        } else { act = fullest(); if (act < 0) { if (lane == 0) *pa.error = 2u; break; } }  // (never: without room for a fresh group something waits)

        if (act == 0 || act == 3) {
            // ================================================================ FRESH
            PT_PHASE(1);
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
            } else {                                                                       // the top of a stack of survivors that holds a full wave;
                // none does (the launch is draining: no camera rays left): the longest walks first, topped up from the shorter ones
                uint32_t t2, t1, t0;
                if (sp2 >= 64u) { t2 = 64u; t1 = 0u; t0 = 0u; }
                else if (sp1 >= 64u) { t2 = 0u; t1 = 64u; t0 = 0u; }
                else if (sp0 >= 64u) { t2 = 0u; t1 = 0u; t0 = 64u; }
                else {
                    t2 = sp2;
                    t1 = sp1 < 64u - t2 ? sp1 : 64u - t2;
                    t0 = sp0 < 64u - t2 - t1 ? sp0 : 64u - t2 - t1;
                }
                valid = lane < t2 + t1 + t0;
                sp2 -= t2; sp1 -= t1; sp0 -= t0;
                const uint32_t off = woff + (lane < t2 ? 2u * (kSFields * STK) + sp2 + lane
                                             : lane < t2 + t1 ? 1u * (kSFields * STK) + sp1 + (lane - t2) : sp0 + (lane - t2 - t1));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, PT_SELF_SCOPE);        // the wave's own stack stores have landed (vmcnt 0) ...
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PT_SELF_SCOPE);        // ... before they are read back through the same L1
                if (valid) {
                    o = mk(ring_ld(off, 0), ring_ld(off, 1), ring_ld(off, 2));
                    d = mk(ring_ld(off, 3), ring_ld(off, 4), ring_ld(off, 5));
                    thr = mk(ring_ld(off, 6), ring_ld(off, 7), ring_ld(off, 8));
                    pv = __float_as_uint(ring_ld(off, 9));
                    level = __float_as_uint(ring_ld(off, 10));
                }
            }
            PT_WSTAT(0, 1); PT_WSTAT(1, __popcll(__ballot(valid)));
            // a slot for every ray of the group
            const u64 vb = __ballot(valid);
            const uint32_t nv = (uint32_t)__popcll(vb);
            uint32_t sid = 0u, pid = 0u;
            if (valid) {
                sid = freel[nfree - 1u - wave_rank(vb)];
                pid = pfree[npfree - 1u - wave_rank(vb)];
                wf[F_OX * R + sid] = o.x; wf[F_OY * R + sid] = o.y; wf[F_OZ * R + sid] = o.z;
                wf[F_DX * R + sid] = d.x; wf[F_DY * R + sid] = d.y; wf[F_DZ * R + sid] = d.z;
#pragma unroll
                for (uint32_t k = 0; k < LW; ++k) wl[(F_L0 + k) * R + sid] = 0xFFFFFFFFu;
                // what only the shading needs waits in the ray's payload record: throughput, pixel word, level, direction
                pay_st(pid, Q_THR, thr.x, thr.y, thr.z, __uint_as_float(pv));
                pay_st(pid, Q_DIR, __uint_as_float(level), d.x, d.y, d.z);
            }
            nfree -= nv; npfree -= nv;
            const CullRay cr = make_cull_ray(o, d);
            uint32_t ncand0 = 0u;
            // one wave-uniform primitive against the group's rays: its own bound per lane, a passing one joins the lane's list
            auto bound_vs_group = [&](uint32_t p) {
                const GeomRec *g = lg + p;
                float tn;
                bool keep;
                const bool pbox = __builtin_amdgcn_readfirstlane(g->type) == 1;
                if (pbox) keep = cull_box(g->bmin, g->bmax, cr, tn);
                else keep = cull_sphere(g->bmin, g->bmax, cr, tn);
                if (keep && valid) {
                    if (ncand0 < kListCap) list_put(sid, ncand0, quant_tn<BIG>(tn, qscale), p, !pbox);
                    ncand0++;
                }
            };
            // ---------------------------------------------------------------- camera rays: ONE cone for the group instead of 64 walks
            // (pt_kernels.hpp, fan_*): lane = one primitive, its bound against the cone; the few it meets are tested by the rays.
            bool fan = false;
            if (PT_FAN && !BIG && act == 3) {       // (wide ids: the pass over every bound is a pass over global memory -- 1.54 -> 1.61 ms/step on 1 024 primitives: the walks stay)
                const int l0 = (int)__builtin_ctzll(vb), l1 = 63 - (int)__builtin_clzll(vb);            // (a FRESH group has a valid lane)
                const f3 e = mk(__shfl(o.x, l0), __shfl(o.y, l0), __shfl(o.z, l0));
                const bool same = !valid || (o.x == e.x && o.y == e.y && o.z == e.z);                    // a common origin (no thin lens)
                const f3 df = mk(__shfl(d.x, l0), __shfl(d.y, l0), __shfl(d.z, l0)), dl = mk(__shfl(d.x, l1), __shfl(d.y, l1), __shfl(d.z, l1));
                const f3 ax = fan_axis(df, dl), nr = fan_normal(df, dl);
                float md = valid ? __builtin_fmaf(d.z, ax.z, __builtin_fmaf(d.y, ax.y, d.x * ax.x)) : 1.0f;
                float mo = valid ? fabsf(__builtin_fmaf(d.z, nr.z, __builtin_fmaf(d.y, nr.y, d.x * nr.x))) : 0.0f;
                if (!(md == md) || !(mo == mo)) md = -1.0f;                                              // a NaN direction: no cone
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) { md = fminf(md, __shfl_xor(md, sft)); mo = fmaxf(mo, __shfl_xor(mo, sft)); }
                FanCone cone;
                fan = fan_finish(e, ax, md, nr, mo, cone) && __ballot(!same) == 0ull;
                if (fan) {
                    for (int base = 0; base < a.G; base += 64) {
                        const int p = base + (int)lane;
                        bool meets = false;
                        if (p < a.G) {
                            const GeomRec *g = lg + p;
                            const int ty = g->type;
                            if (ty == 0 || ty == 1) meets = fan_meets(g->bmin, g->bmax, ty == 0, cone);
                        }
                        u64 m = __ballot(meets);
                        while (m) {
                            const uint32_t j = (uint32_t)__builtin_ctzll(m);
                            m &= m - 1ull;
                            bound_vs_group((uint32_t)base + j);
                        }
                    }
                }
            }
            // ---------------------------------------------------------------- the big primitives: every ray, wave-uniform index
            if (!fan)
                for (int k = 0; k < nbig; ++k) bound_vs_group((uint32_t)__builtin_amdgcn_readfirstlane((int)ld_big(k)));
            PT_WSTAT(25, (unsigned long long)nbig * nv);
            // ---------------------------------------------------------------- walk set-up
            // Walked: finite rays that start within `reach` of the grid (the others overflow their list on purpose).
            const GridArgs &lga = *reinterpret_cast<const GridArgs *>(ctrl + 100);
            const bool sane = grid_walk_sane(lga, o, d);
            if (valid && !sane) ncand0 = kListCap + 1u;
            PT_WSTAT(24, __popcll(__ballot(valid && !sane)));
            if (valid) wl[F_META * R + sid] = ncand0;                                      // candidate count while the pairs are worked off
            GridWalk gw;
            gw.walking = false;
            if (!fan) gw = grid_walk_begin(lga, o, d, cr.inv, valid && sane);
            const uint32_t max_trips = (uint32_t)__builtin_amdgcn_readfirstlane(lga.n[0] + lga.n[1] + lga.n[2]) + 2u;     // a walk takes at most n_x + n_y + n_z - 2 steps
            uint32_t trips = 0u;
            // ---------------------------------------------------------------- WALK / CELLS / BOUNDS under one dispatcher
            uint32_t nA = 0u, nCb = 0u, nCs = 0u;
            bool walk_left = true;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                         // the slots are written before any pair lane reads them
            __builtin_amdgcn_wave_barrier();
            for (;;) {
                int op;                                                                    // 0 walk, 1 CELLS, 3 BOUNDS cubes, 4 BOUNDS spheres
                if (nCb >= 64u) op = 3;
                else if (nCs >= 64u) op = 4;
                else if (nA >= 64u) op = 1;
                else if (walk_left) op = 0;
                else if (nA) op = 1;
                else if (nCb) op = 3;
                else if (nCs) op = 4;
                else break;
                if (op == 0) {
                    // ------------------------------------------------------------ WALK: one cell per trip, until 64 entries wait
                    PT_PHASE(2);
                    for (;;) {
                        const u64 wb = __ballot(gw.walking);
                        if (wb == 0ull || trips >= max_trips) { walk_left = false; break; }
                        trips++;
                        PT_WSTAT(2, 1); PT_WSTAT(3, __popcll(wb));
#if PT_WALK_FLAT
                        // every lane reads and steps, walking or not: a lane that has left the grid holds a meaningless cell index, and an LDS
                        // read beyond the block's allocation returns zero -- its `emit` is masked below
                        uint32_t rec = BIG ? ld_cell(gw.walking ? grid_walk_cell(gw) : 0u) : cells[grid_walk_cell(gw)];
                        const bool emit = gw.walking && (BIG ? rec != 0xFFFFFFFFu : (rec >> 16) != 0u);
#else
                        uint32_t rec = BIG ? 0xFFFFFFFFu : 0u;
                        if (gw.walking) rec = ld_cell(grid_walk_cell(gw));
                        const bool emit = BIG ? rec != 0xFFFFFFFFu : (rec >> 16) != 0u;
#endif
                        const u64 eb = __ballot(emit);
                        if (emit) bufA[nA + wave_rank(eb)] = sid | ((BIG ? rec : (rec & 0xFFFFu)) << Fm::kRefShift) | (gw.emask << Fm::kEmaskShift);  // the cell's first reference
                        nA += (uint32_t)__popcll(eb);
                        PT_WSTAT(4, __popcll(eb));
#if PT_WALK_FLAT
                        grid_walk_step(gw);
#else
                        if (gw.walking) grid_walk_step(gw);
#endif
                        if (nA >= 64u) break;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else if (op == 1) {
                    // ------------------------------------------------------------ CELLS: the top 64 (ray, reference) entries, lane = one entry;
                    // a new primitive goes on to BOUNDS, and unless the reference is its cell's last, the entry returns for the next one
                    PT_PHASE(3);
                    const uint32_t n = nA < 64u ? nA : 64u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint32_t e = 0u, ref = BIG ? 0x40000000u : 0x4000u;
                    const bool evalid = lane < n;
                    if (evalid) {
                        e = bufA[nA - n + lane];
                        if constexpr (BIG) ref = ld_ref32((e >> Fm::kRefShift) & Fm::kRefMask);
                        else ref = (uint32_t)refs[(e >> Fm::kRefShift) & Fm::kRefMask];
                    }
                    nA -= n;
                    PT_WSTAT(5, 1); PT_WSTAT(6, n);
                    const uint32_t rflags = BIG ? ref >> 24 : ref >> 8, rprim = BIG ? (ref & 0xFFFFFFu) : (ref & 0xFFu);
                    const bool isnew = evalid && grid_flags_new(rflags, e >> Fm::kEmaskShift);
                    const bool sph = (rflags & 0x80u) != 0u;
                    const bool again = (rflags & 0x40u) == 0u;                             // (invalid lanes: `last` is set)
                    const u64 bb = __ballot(isnew && !sph), sb = __ballot(isnew && sph), gb = __ballot(again);
                    __builtin_amdgcn_wave_barrier();                                       // every entry is read before any is overwritten
                    if (isnew) {
                        const uint32_t pos = sph ? kBufC - 1u - (nCs + wave_rank(sb)) : nCb + wave_rank(bb);
                        bufC[pos] = (e & ((1u << Fm::kSidBits) - 1u)) | (rprim << 8);
                    }
                    if (again) bufA[nA + wave_rank(gb)] = e + (1u << Fm::kRefShift);
                    nCb += (uint32_t)__popcll(bb); nCs += (uint32_t)__popcll(sb); nA += (uint32_t)__popcll(gb);
                    PT_WSTAT(9, __popcll(bb) + __popcll(sb));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    // ------------------------------------------------------------ BOUNDS: the top 64 (ray, primitive) entries of one type
                    PT_PHASE(4);
                    const bool isb = op == 3;
                    const uint32_t have = isb ? nCb : nCs;
                    const uint32_t n = have < 64u ? have : 64u;
                    const bool pvalid = lane < n;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint32_t psid = 0u, pp = 0u;
                    f3 po = mk(0, 0, 0), pd = mk(0, 0, 1);
                    if (pvalid) {
                        const uint32_t e = bufC[isb ? have - n + lane : kBufC - 1u - (have - n + lane)];
                        psid = e & 0xFFu; pp = e >> 8;
                        po = mk(wf[F_OX * R + psid], wf[F_OY * R + psid], wf[F_OZ * R + psid]);
                        pd = mk(wf[F_DX * R + psid], wf[F_DY * R + psid], wf[F_DZ * R + psid]);
                    }
                    if (isb) nCb -= n; else nCs -= n;
                    PT_WSTAT(isb ? 10 : 12, 1); PT_WSTAT(isb ? 11 : 13, n);
                    const GeomRec *g = lg + pp;                                            // per-lane gather from the geometry table
                    float tn = 0.0f;
                    bool keep;
                    if (isb) { const CullRay pr = make_cull_ray(po, pd); keep = cull_box(g->bmin, g->bmax, pr, tn); }
                    else { CullRay pr; pr.o = po; pr.d = pd; pr.inv = pd; pr.noi = pd; keep = cull_sphere(g->bmin, g->bmax, pr, tn); }
                    if (pvalid && keep) {
                        const uint32_t pos = atomicAdd(&wl[F_META * R + psid], 1u);
#ifdef PT_CULL_STATS
                        atomicAdd(&g_wstats[14], 1ull);
#endif
                        if (pos < kListCap) list_put(psid, pos, quant_tn<BIG>(tn, qscale), pp, !isb);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---------------------------------------------------------------- SELECT: nearest candidate first
            PT_PHASE(5);
            bool queued = false, tobox = false;
            const uint32_t ncand = valid ? wl[F_META * R + sid] : 0u;
            // Rays whose list overflowed (rare: 2 in 10 000 on configs[3]) or that were not walked take the reference loop itself; its
            // winner is then confirmed by one exact test in a TEST group.  A few of them: one ray at a time on the WHOLE wave, every
            // lane testing G / 64 primitives + a min-reduction with the reference's tie rule -- a lane alone would keep the wave
            // waiting for G exact tests.  Many: the per-lane loop.
            int ovhit = -1;
            {
                const bool ov = ncand > kListCap;
                const u64 ovb = __ballot(ov);
                PT_WSTAT(15, __popcll(ovb));
                if (ovb) {
                    if (__popcll(ovb) >= 12) {
                        if (ov) { float tb2; f3 P, N; ovhit = nearest_hit(lg, a.G, o, d, tb2, P, N); }
                    } else {
                        u64 m = ovb;
                        while (m) {
                            const int src = (int)__builtin_ctzll(m);
                            m &= m - 1ull;
                            const f3 oo = mk(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                            const f3 dd = mk(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                            float bd = kInf;
                            int bh = -1;
                            if constexpr (!BIG) {                                   // up to 256 primitives: four trips
                                for (int k = (int)lane; k < a.G; k += 64) {
                                    const GeomRec *g = lg + k;
                                    const int ty = g->type;
                                    const bool pb = ty == 1, psph = ty == 0;
                                    float depth = -1.0f;
                                    f3 P, N;
                                    if (__any(pb)) { if (pb) depth = box_test(g->inv, g->xf, g->inside_hits, oo, dd, P, N); }
                                    if (__any(psph)) { if (psph) depth = sphere_test(g->inv, g->xf, oo, dd, P, N); }
                                    if (depth > -PT_EPSILON && (depth < bd || (depth == bd && k < bh))) { bd = depth; bh = k; }
                                }
                            } else {
                            // first every primitive's BOUND against this ray (lane = primitive; the pair buffers are empty here: the
                            // second one takes the ids, cubes from the bottom, spheres from the top), then the exact tests of the few that
                            // pass on dense lanes.  A ray whose bounds nearly all pass (non-finite: false comparisons keep everything)
                            // takes the plain loop over all primitives.
                            const CullRay pr = make_cull_ray(oo, dd);
                            uint32_t nb2 = 0u, ns2 = 0u;
                            bool fits = true;
                            for (int base = 0; base < a.G && fits; base += 64) {
                                const int k = base + (int)lane;
                                bool pb = false, psph = false;
                                if (k < a.G) {
                                    const GeomRec *g = lg + k;
                                    const int ty = g->type;
                                    float tn;
                                    if (ty == 1) pb = cull_box(g->bmin, g->bmax, pr, tn);
                                    else if (ty == 0) psph = cull_sphere(g->bmin, g->bmax, pr, tn);
                                }
                                const u64 cb2 = __ballot(pb), sb2 = __ballot(psph);
                                const uint32_t cn = (uint32_t)__popcll(cb2), sn = (uint32_t)__popcll(sb2);
                                if (nb2 + ns2 + cn + sn > kBufC) { fits = false; break; }
                                if (pb) bufC[nb2 + wave_rank(cb2)] = (uint32_t)k;
                                if (psph) bufC[kBufC - 1u - (ns2 + wave_rank(sb2))] = (uint32_t)k;
                                nb2 += cn; ns2 += sn;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if (fits) {
                                for (uint32_t i = lane; i < nb2; i += 64u) {
                                    const int k = (int)bufC[i];
                                    const GeomRec *g = lg + k;
                                    f3 P, N;
                                    const float depth = box_test(g->inv, g->xf, g->inside_hits, oo, dd, P, N);
                                    if (depth > -PT_EPSILON && (depth < bd || (depth == bd && k < bh))) { bd = depth; bh = k; }
                                }
                                for (uint32_t i = lane; i < ns2; i += 64u) {
                                    const int k = (int)bufC[kBufC - 1u - i];
                                    const GeomRec *g = lg + k;
                                    f3 P, N;
                                    const float depth = sphere_test(g->inv, g->xf, oo, dd, P, N);
                                    if (depth > -PT_EPSILON && (depth < bd || (depth == bd && k < bh))) { bd = depth; bh = k; }
                                }
                            } else {
                                for (int k = (int)lane; k < a.G; k += 64) {
                                    const GeomRec *g = lg + k;
                                    const int ty = g->type;
                                    const bool pb = ty == 1, psph = ty == 0;
                                    float depth = -1.0f;
                                    f3 P, N;
                                    if (__any(pb)) { if (pb) depth = box_test(g->inv, g->xf, g->inside_hits, oo, dd, P, N); }
                                    if (__any(psph)) { if (psph) depth = sphere_test(g->inv, g->xf, oo, dd, P, N); }
                                    if (depth > -PT_EPSILON && (depth < bd || (depth == bd && k < bh))) { bd = depth; bh = k; }
                                }
                            }
                            __builtin_amdgcn_wave_barrier();                      // the ids are read before the next ray's overwrite them
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
                        tobox = lg[first_id].type == 1;
#pragma unroll
                        for (uint32_t k = 0; k < LW; ++k) wl[(F_L0 + k) * R + sid] = 0xFFFFFFFFu;
                    }
                } else if (cnt != 0u) {
                    uint32_t L[LW];
#pragma unroll
                    for (uint32_t k = 0; k < LW; ++k) L[k] = wl[(F_L0 + k) * R + sid];
                    uint32_t pos;
                    const uint32_t key = select_next<BIG>(L, Fm::kQMax, pos);
                    queued = true;
                    if constexpr (BIG) { first_id = key & kBigIdMask; tobox = (key & kBigSphere) == 0u; }
                    else { first_id = key & 0xFFu; tobox = lg[first_id].type == 1; }
                    list_clear(sid, pos, L);                                               // the chosen entry leaves the list
                }
                if (queued) {
                    if constexpr (BIG) { wl[F_META * R + sid] = first_id | (pid << 24); wl[F_HIT * R + sid] = 0u; }     // current candidate, no hit yet
                    else wl[F_META * R + sid] = first_id | (pid << 19);
                    wf[F_BEST * R + sid] = kInf;
                }
            }
            {
                const u64 bb = __ballot(queued && tobox), sb = __ballot(queued && !tobox), fb = __ballot(valid && !queued);
                if (queued) {
                    const uint32_t pos = tobox ? nbox + wave_rank(bb) : (uint32_t)R - 1u - (nsph + wave_rank(sb));
                    xstack[pos] = (unsigned char)sid;
                }
                if (valid && !queued) {                                                    // no candidate at all: the ray leaves the scene
                    freel[nfree + wave_rank(fb)] = (unsigned char)sid;
                    pfree[npfree + wave_rank(fb)] = (unsigned char)pid;
                }
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                nfree += (uint32_t)__popcll(fb); npfree += (uint32_t)__popcll(fb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }

        if (act == 1 || act == 2) {
            // ================================================================ TEST (one type per group, any levels)
            PT_PHASE(6);
            const bool isb = act == 1;
            const uint32_t have = isb ? nbox : nsph;
            const uint32_t cnt = have < 64u ? have : 64u;
            const bool valid = lane < cnt;
            const uint32_t qpos = isb ? (have - cnt + lane) : ((uint32_t)R - 1u - (have - cnt + lane));
            if (isb) nbox -= cnt; else nsph -= cnt;
            PT_WSTAT(isb ? 16 : 18, 1); PT_WSTAT(isb ? 17 : 19, cnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t sid = 0u, meta = 0u, hitw = 0u;
            uint32_t L[LW];
#pragma unroll
            for (uint32_t k = 0; k < LW; ++k) L[k] = 0xFFFFFFFFu;
            f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            float best = kInf;
            if (valid) {
                sid = xstack[qpos];
                o = mk(wf[F_OX * R + sid], wf[F_OY * R + sid], wf[F_OZ * R + sid]);
                d = mk(wf[F_DX * R + sid], wf[F_DY * R + sid], wf[F_DZ * R + sid]);
                meta = wl[F_META * R + sid];
                if constexpr (BIG) hitw = wl[F_HIT * R + sid];
                best = wf[F_BEST * R + sid];
#pragma unroll
                for (uint32_t k = 0; k < LW; ++k) L[k] = wl[(F_L0 + k) * R + sid];
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t j = BIG ? (meta & kBigIdMask) : (meta & 0xFFu), pid = BIG ? (meta >> 24) : ((meta >> 19) & 0xFFu);
            bool has_hit = BIG ? (hitw >> 31) != 0u : (meta & kMetaHasHit) != 0u;
            uint32_t hit = BIG ? (hitw & kBigIdMask) : ((meta >> 8) & 0xFFu);
            bool hit_box = BIG ? ((hitw >> 30) & 1u) != 0u : false;                   // (narrow: looked up in the LDS table)
            int face = (int)((meta >> (BIG ? 21 : 16)) & 7u) - 1;
            bool won = false;
            f3 P = mk(0, 0, 0), N = mk(0, 0, 0);
            {
                const GeomRec *gr = lg + j;                                   // per-lane gather from the LDS table
                float depth = -1.0f;
                int fc = -1;
                if (valid) {
                    if (isb) depth = box_test_face(gr->inv, gr->xf, gr->inside_hits, o, d, P, fc);
                    else depth = sphere_test(gr->inv, gr->xf, o, d, P, N);
                }
                // nearest-hit update of the reference loop: first strictly nearer wins, ties to the lower index
                won = valid && depth > -PT_EPSILON && (has_hit ? (depth < best || (depth == best && j < hit)) : depth < kInf);
                if (won) { best = depth; hit = j; face = fc; has_hit = true; hit_box = isb; }
            }
            // the new best hit's point and normal wait in the payload record (an earlier winner's stay there otherwise)
            if (won) {
                pay_st(pid, Q_HIT, P.x, P.y, P.z, __uint_as_float(hit | ((uint32_t)(face + 1) << (BIG ? 24 : 8))));
                if (!isb) pay_st(pid, Q_NRM, N.x, N.y, N.z, 0.0f);
            }
            // the next candidate that could still win or tie: key distance not beyond the best hit (conservative: one step of slack)
            PT_PHASE(7);
            uint32_t npos = 0u;
            uint32_t qmax = Fm::kQMax;
            if (has_hit) {
                const float lim = fminf((best + slack_max) * qscale, (float)(Fm::kQMax - 1u));
                qmax = (uint32_t)lim + 1u;
            }
            const uint32_t nkey = valid ? select_next<BIG>(L, qmax, npos) : Fm::kEmpty;
            const bool more = nkey != Fm::kEmpty;
            const bool done = valid && !more;
            bool nbx = false;
            if (more) {
                const uint32_t nid = BIG ? (nkey & kBigIdMask) : (nkey & 0xFFu);
                list_clear(sid, npos, L);
                if constexpr (BIG) {
                    wl[F_META * R + sid] = nid | ((uint32_t)(face + 1) << 21) | (pid << 24);
                    wl[F_HIT * R + sid] = hit | (hit_box ? 1u << 30 : 0u) | (has_hit ? 1u << 31 : 0u);
                    nbx = (nkey & kBigSphere) == 0u;
                } else {
                    wl[F_META * R + sid] = nid | (hit << 8) | ((uint32_t)(face + 1) << 16) | (pid << 19) | (has_hit ? kMetaHasHit : 0u);
                    nbx = lg[nid].type == 1;
                }
                wf[F_BEST * R + sid] = best;
            }
            // a finished ray leaves its slot: with a hit, its payload record waits for the shading of that TYPE of primitive
            const bool toshade = done && has_hit;
            bool shb = false;
            if (toshade) {
                if constexpr (BIG) shb = hit_box; else shb = lg[hit].type == 1;
            }
            PT_WSTAT(20, __popcll(__ballot(toshade))); PT_WSTAT(21, __popcll(__ballot(done)));
            PT_WSTAT(22, __popcll(__ballot(more))); PT_WSTAT(23, __popcll(__ballot(more && won)));
            PT_PHASE(9);
            {
                const u64 bb = __ballot(more && nbx), sb = __ballot(more && !nbx), fb = __ballot(done);
                const u64 hb = __ballot(toshade && shb), hs = __ballot(toshade && !shb), pb = __ballot(done && !has_hit);
                if (more) {
                    const uint32_t pos = nbx ? nbox + wave_rank(bb) : (uint32_t)R - 1u - (nsph + wave_rank(sb));
                    xstack[pos] = (unsigned char)sid;
                }
                if (done) freel[nfree + wave_rank(fb)] = (unsigned char)sid;
                if (toshade) {
                    const uint32_t pos = shb ? nshb + wave_rank(hb) : (uint32_t)NP - 1u - (nshs + wave_rank(hs));
                    shstack[pos] = (unsigned char)pid;
                }
                if (done && !has_hit) pfree[npfree + wave_rank(pb)] = (unsigned char)pid;
                nbox += (uint32_t)__popcll(bb);
                nsph += (uint32_t)__popcll(sb);
                nfree += (uint32_t)__popcll(fb);
                nshb += (uint32_t)__popcll(hb); nshs += (uint32_t)__popcll(hs); npfree += (uint32_t)__popcll(pb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }

        // ==================================================================== SHADE (hits on one type of primitive, any levels)
        PT_PHASE(8);
        {
            const bool hbx = act == 4;
            const uint32_t have = hbx ? nshb : nshs;
            const uint32_t cnt = have < 64u ? have : 64u;
            const bool valid = lane < cnt;
            if (hbx) nshb -= cnt; else nshs -= cnt;
            PT_WSTAT(hbx ? 26 : 28, 1); PT_WSTAT(hbx ? 27 : 29, cnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, PT_SELF_SCOPE);            // payload stores of earlier groups have landed ...
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PT_SELF_SCOPE);            // ... before they are read back through the same L1
            uint32_t pid = 0u, pv = 0u, level = 0u, hf = 0u;
            f3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(0, 0, 0), P = mk(0, 0, 0), N = mk(0, 0, 0);
            if (valid) {
                pid = shstack[hbx ? have - cnt + lane : (uint32_t)NP - 1u - (have - cnt + lane)];
                const u32x4 qa = pay_ld(pid, Q_THR), qb = pay_ld(pid, Q_DIR), qc = pay_ld(pid, Q_HIT);
                thr = mk(__uint_as_float(qa.x), __uint_as_float(qa.y), __uint_as_float(qa.z)); pv = qa.w;
                level = qb.x; d = mk(__uint_as_float(qb.y), __uint_as_float(qb.z), __uint_as_float(qb.w));
                P = mk(__uint_as_float(qc.x), __uint_as_float(qc.y), __uint_as_float(qc.z)); hf = qc.w;
                if (!hbx) { const u32x4 qd = pay_ld(pid, Q_NRM); N = mk(__uint_as_float(qd.x), __uint_as_float(qd.y), __uint_as_float(qd.z)); }
            }
            const uint32_t hit = BIG ? (hf & 0xFFFFFFu) : (hf & 0xFFu);
            const int face = (int)((hf >> (BIG ? 24 : 8)) & 7u) - 1;
            bool alive = false;
            if (valid) {
                const uint32_t slot = a.batch > 1u ? pv >> 24 : 0u, pixel = pv & a.pix_mask;
                const MatRec m = lm[lg[hit].mat];
                if (level + 1u >= D && !(m.emittance > 0.0f)) {
                    alive = true;                                             // depth exhausted: alive, contributes 0
                } else {
                    const uint32_t iteration = a.iteration + slot;
                    uint32_t st = lcg_seed(stream_seed(pixel, iteration, 1u + level));
                    st = lcg_next(st); const float u_sel = u01(st);
                    st = lcg_next(st); const float xi1 = u01(st);
                    st = lcg_next(st); const float xi2 = u01(st);
                    f3 Lr = mk(0.0f, 0.0f, 0.0f);
                    int code;
                    if (hbx) code = scatter_box(m, P, face, frames + 3 * hit, u_sel, xi1, xi2, o, d, thr, Lr);
                    else code = scatter(m, P, N, u_sel, xi1, xi2, o, d, thr, Lr);
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
            // ---------------------------------------------------------------- the payload records are free again; survivors wait for their next bounce
            if (valid) pfree[npfree + lane] = (unsigned char)pid;
            npfree += cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool onward = alive && level + 1u < D;
            if (pa.tap_level != 0u) {                                         // parity hook: the rays entering bounce tap_level leave here
                const bool tapped = onward && level + 1u == pa.tap_level;
                const u64 tb2 = __ballot(tapped);
                if (tb2) {
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(pa.tap_count, (uint32_t)__popcll(tb2));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (tapped) {
                        float *t = pa.tap + base + wave_rank(tb2);
                        const size_t cap = pa.tap_cap;
                        t[0 * cap] = o.x; t[1 * cap] = o.y; t[2 * cap] = o.z; t[3 * cap] = d.x; t[4 * cap] = d.y; t[5 * cap] = d.z;
                        t[6 * cap] = thr.x; t[7 * cap] = thr.y; t[8 * cap] = thr.z; t[9 * cap] = __uint_as_float(pv);
                    }
                }
                onward = onward && !tapped;
            }
            if (__any(onward)) {
                // the stack by the walk ahead of the scattered ray (rays the kernel will not walk: the short one)
                const GridArgs &lga = *reinterpret_cast<const GridArgs *>(ctrl + 100);
                uint32_t wlen = 0u;
                if (onward && grid_walk_sane(lga, o, d)) wlen = grid_walk_length(lga, o, d, mk(guarded_rcp(d.x), guarded_rcp(d.y), guarded_rcp(d.z)));
                const uint32_t bin = wlen <= lga.bin1 ? 0u : wlen <= lga.bin2 ? 1u : 2u;
                const u64 b0 = __ballot(onward && bin == 0u), b1 = __ballot(onward && bin == 1u), b2 = __ballot(onward && bin == 2u);
                const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2);
                if (sp0 + n0 > STK || sp1 + n1 > STK || sp2 + n2 > STK) { if (lane == 0) *pa.error = 2u; }      // never: see STK
                else {
                    if (onward) {
                        const uint32_t at = bin == 0u ? sp0 + wave_rank(b0) : bin == 1u ? sp1 + wave_rank(b1) : sp2 + wave_rank(b2);
                        const uint32_t off = woff + bin * (kSFields * STK) + at;
                        ring_st(off, 0, o.x); ring_st(off, 1, o.y); ring_st(off, 2, o.z);
                        ring_st(off, 3, d.x); ring_st(off, 4, d.y); ring_st(off, 5, d.z);
                        ring_st(off, 6, thr.x); ring_st(off, 7, thr.y); ring_st(off, 8, thr.z);
                        ring_st(off, 9, __uint_as_float(pv));
                        ring_st(off, 10, __uint_as_float(level + 1u));
                    }
                    sp0 += n0; sp1 += n1; sp2 += n2;
                }
            }
        }
    }

#ifdef PT_CULL_STATS
    PT_PHASE(0);
    if (lane == 0) for (int i = 0; i < 10; ++i) atomicAdd(&g_phase_cycles[i], ph_acc[i]);
#endif
    for (int sft = 32; sft > 0; sft >>= 1) emitted += __shfl_down(emitted, sft);
    if (lane == 0 && emitted) atomicAdd(&ctrl[1], emitted);
    __syncthreads();
    if (threadIdx.x == 0 && ctrl[1]) atomicAdd(&a.sync->emitted, (u64)ctrl[1]);
    if (threadIdx.x >= 1 && threadIdx.x < 65 && lsurv[threadIdx.x]) atomicAdd(&bank[threadIdx.x], lsurv[threadIdx.x]);
}

// ------------------------------------------------------------------ host side ---------
namespace {
struct WideShape { int waves, slots, payload; bool big; };
// shape -> block.  One block per CU; the waves share the CU's LDS: tables + grid + waves x (slot fields x slots + the byte lists + the
// two pair buffers).  0: narrow default; 1: narrow, what 0 falls back to when a large grid leaves less room; 2, 3: wide ids (more
// than 256 primitives: 17 slot fields, no geometry table in LDS) with more slots or more room for the grid
constexpr WideShape kShapes[4] = {{16, 112, 240, false}, {16, 80, 208, false}, {16, 96, 224, true}, {16, 80, 208, true}};

const void *wide_fn_of(int v) {
    switch (v) {
    case 1: return reinterpret_cast<const void *>(&k_path_w<16, 80, 208, false>);
    case 2: return reinterpret_cast<const void *>(&k_path_w<16, 96, 224, true>);
    case 3: return reinterpret_cast<const void *>(&k_path_w<16, 80, 208, true>);
    default: return reinterpret_cast<const void *>(&k_path_w<16, 112, 240, false>);
    }
}
uint32_t shape_lds(const WideShape &s, int G, int M, uint32_t grid_bytes) {
    const uint32_t fields = s.big ? 17u : 12u;
    return tables_bytes(G, M, !s.big) + grid_bytes +
           (uint32_t)s.waves * (fields * (uint32_t)s.slots + (2u * (uint32_t)s.slots + 2u * (uint32_t)s.payload) / 4u + kBufA + kBufC) * 4u;
}
}  // namespace

// LDS a shape needs beside `grid_bytes` of grid (0: the grid stays in global memory; wide shapes only)
uint32_t wide_lds_bytes(int shape, int G, int M, uint32_t grid_bytes) { return shape_lds(kShapes[shape & 3], G, M, grid_bytes); }

hipError_t wide_setup(int shape, int G, int M, uint32_t grid_bytes, WideLayout *out) {
    const WideShape s = kShapes[shape & 3];
    const uint32_t lds = shape_lds(s, G, M, grid_bytes);
    out->waves_per_block = (uint32_t)s.waves;
    out->slots_per_wave = (uint32_t)s.slots;
    out->payload_per_wave = (uint32_t)s.payload;
    out->stack_slots = (uint32_t)((s.payload + 64 * (int)kWalkBins + 63) / 64 * 64);
    out->lds_bytes = lds;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;        // no room for this shape (the caller tries the next)
    return hipFuncSetAttribute(wide_fn_of(shape & 3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void wide_launch(int shape, int grid, uint32_t lds, hipStream_t st, const SegArgs &a, const PathArgs &pa, const GridArgs &ga,
                 const GeomRec *g, const MatRec *m, const FaceFrame *frames) {
    switch (shape & 3) {
    case 1: hipLaunchKernelGGL((k_path_w<16, 80, 208, false>), dim3(grid), dim3(16 * 64), lds, st, a, pa, ga, g, m, frames); break;
    case 2: hipLaunchKernelGGL((k_path_w<16, 96, 224, true>), dim3(grid), dim3(16 * 64), lds, st, a, pa, ga, g, m, frames); break;
    case 3: hipLaunchKernelGGL((k_path_w<16, 80, 208, true>), dim3(grid), dim3(16 * 64), lds, st, a, pa, ga, g, m, frames); break;
    default: hipLaunchKernelGGL((k_path_w<16, 112, 240, false>), dim3(grid), dim3(16 * 64), lds, st, a, pa, ga, g, m, frames); break;
    }
}

#ifdef PT_CULL_STATS
void phase_cycles_wide(unsigned long long *out16) {
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long));
}
void stats_wide(unsigned long long *out32) {
    (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wstats), 32 * sizeof(unsigned long long));
}
void cull_stats_wide(unsigned long long *) {}
#endif

}  // namespace ptk
