// lmaze_step.hip -- the hot path: one kernel per step() of N independent mazes, gfx950.
//
// Replaces the body of the reference's step() (gym_lmaze/envs/lmaze_env.py:146-237,
// lmaze_env_v3.py:220-402): action decode, wall-collision check, position update,
// reward/done, and the full re-render of the observation planes.
//
// Shape of the work: per env 37 B of state traffic and 4*G*G B of
// observation written -- an HBM-write-bound stream with integer indexing, no MFMA.
//   phase 1  one lane per env: coalesced SoA loads, transition against the layout held in
//            LDS, coalesced SoA stores; the new ball cell goes to LDS.
//   phase 2  the workgroup's envs own one contiguous [envs*G*G] dword range of obs; lanes
//            stripe across it with 16-byte stores (1 KiB per wave instruction), so the
//            write is perfectly coalesced whatever G is.  Shared layout: the ball-free
//            pattern of a 16-byte-periodic group of envs is precomputed in LDS, so a
//            store costs one ds_read_b128 + a few compares.  Per-env layouts: the
//            workgroup's layouts are tiled into LDS first (one coalesced read), and both
//            the collision check and the render read them from there.
// Options folded into the same kernels (wave-uniform branches, no extra traffic):
//   auto_reset  an env whose done flag is set on entry is first re-placed exactly as
//               lmaze_reset(mask = done) would (reference reset(), lmaze_env.py:64-110)
//   mask        (observe only) re-render just the envs a masked reset touched
#include <cstdio>

#include <cmath>
#include <cstdlib>

#include "lmaze_common.h"
#include "lmaze_policy_table.h"

namespace lmaze {

// Obs buffers larger than this are written with non-temporal stores: they cannot stay in the
// 256 MiB Infinity Cache anyway, and streaming them past L2 measured 0-7 % faster on MI355X;
// smaller buffers keep plain stores so the consumer of the observation finds them on-die.
static constexpr size_t kNonTemporalObsBytes = (size_t)192 << 20;

template <bool NT, int BITS = 0>
__device__ __forceinline__ void store16(int4* p, const int4& v) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    if (NT) {
        v4i t = {v.x, v.y, v.z, v.w};
        stream_store16<BITS>(reinterpret_cast<v4i*>(p), t);
    } else {
        *p = v;
    }
}

// goalCount += 1 (v0:195) as an atomic without return: a plain read-modify-write makes the wave wait for the load --
// and with it for every load issued before, i.e. for the NEXT chunk's prefetched inputs -- whenever one of its 64 envs
// reaches the goal (about every second wave of a random rollout).  One lane per env, so the atomicity itself is unused.
__device__ __forceinline__ void count_goal(int32_t* p) { __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One env's inputs, loaded into registers BEFORE the workgroup's LDS set-up and first barrier so
// that the two global round trips (layout, state) overlap instead of chaining: at launch-bound
// batch sizes (65 536 x 8x8) the kernel is nothing but that latency chain.
struct EnvIn {
    int2 b, g;
    int act, sc, was_done;
    float r;
};

template <int VARIANT, bool DO_STEP>
__device__ __forceinline__ EnvIn load_env(const StepArgs& a, int64_t e) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    EnvIn in;
    in.b = a.ball[e];
    in.g = make_int2(-1, -1);
    if (V3) in.g = a.goal[e];
    in.act = 0; in.sc = 0; in.was_done = 0; in.r = 0.0f;
    if (DO_STEP) {
        in.act = a.action[e];
        in.sc = a.step_count[e];
        if (!V3) in.r = a.reward[e];
        // branch-free (a load under a branch is waited for where the branch ends, in front of whatever else is in flight):
        // without the fused reset the lane reads byte 0 of the layout instead -- one cached address, no traffic
        const uint8_t* dp = a.auto_reset ? a.done + e : a.layout;
        // ... and the byte is kept as loaded: every use is `a.auto_reset && in.was_done`.  A select here is a USE of the
        // loaded value inside the caller's `if (tid < nb)`: the wave then waits for its state loads before the set-up's
        // layout loads are even issued -- two global round trips in series at the head of every workgroup.
        in.was_done = *dp;
    }
    return in;
}

// one env of phase 1 (layout in LDS); returns the env's ball (and goal) cell index for phase 2
// `place(draw, ball_cell, goal_cell)` = the fused reset's placement over the accepted cells (a list in LDS or bit masks)
template <int VARIANT, bool DO_STEP, class Place>
__device__ __forceinline__ void env_phase1(const StepArgs& a, const uint8_t* lay, int G, int64_t e, EnvIn in,
                                           Place place, uint64_t epoch, int& ball_cell, int& goal_cell) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    int2 b = in.b, g = in.g;
    if (DO_STEP) {
        int sc_in = in.sc;
        float r_in = in.r;
        if (a.auto_reset && in.was_done) {  // reference reset(): placement + zeroed counters
            int bc, gc;
            place(env_draw(a.seed, epoch, a.env_base + e), bc, gc);
            if (bc >= 0) b = make_int2(bc / G, bc % G);
            if (V3 && gc >= 0) {
                g = make_int2(gc / G, gc % G);
                a.goal_rw[e] = g;
            }
            sc_in = 0;      // v0:110
            r_in = -0.0f;   // v0:109
        }
        int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
        const int sc = sc_in + 1;  // v0:151, v3:225
        int ox, oy;
        decode_action(in.act, ox, oy);
        const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
        float r;
        bool dn;
        if (transition_rule<VARIANT>(a, lay[tx * G + ty], ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn) && a.goal_count)
            count_goal(a.goal_count + e);
        a.ball[e] = make_int2(bx, by);
        a.step_count[e] = sc;
        a.reward[e] = r;
        a.done[e] = dn ? 1 : 0;
        b = make_int2(bx, by);
    } else {
        b = make_int2(clampi(b.x, 0, G - 1), clampi(b.y, 0, G - 1));
    }
    ball_cell = b.x * G + b.y;
    goal_cell = (V3 && g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
}

// ------------------------------------------------------------------------------------
// Shared layout.  GT = G known at compile time (0: read it from the args);
// EPB = envs per workgroup (multiple of 4); NT = non-temporal obs stores.
// ------------------------------------------------------------------------------------
template <int GT, int VARIANT, bool DO_STEP, int EPB, bool NT>
__global__ __launch_bounds__(LMAZE_BLOCK) void step_shared_kernel(const StepArgs a) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    static_assert(EPB % 4 == 0 && EPB <= LMAZE_BLOCK, "group alignment; one lane per env");
    const int G = GT ? GT : a.grid;
    const int CELLS = G * G;
    // envs per 16-byte period of the obs stream: G even -> an env is a whole number of
    // 16-byte stores; G odd -> four envs are (G*G ints each, 4*G*G ints = G*G stores)
    const int GRP = (GT != 0 && (G & 1)) ? 4 : 1;
    const int PAT = (GT != 0) ? GRP * CELLS : CELLS;  // dwords of ball-free pattern kept in LDS

    extern __shared__ int4 lds4[];
    int* pat = reinterpret_cast<int*>(lds4);
    int* ballflat = pat + PAT;
    int* goalflat = ballflat + EPB;
    int* maskflag = goalflat + EPB;
    uint8_t* lay = reinterpret_cast<uint8_t*>(maskflag + EPB);
    uint16_t* spawn = reinterpret_cast<uint16_t*>(lay + ((CELLS + 15) & ~15));
    // v3, compile-time G: where the balls and goals are, as bit strings -- bit f of a group's string = dword f of the
    // group's GRP*G*G output dwords holds one -- so that a 16-byte store ORs two nibbles into its pattern instead of
    // comparing eight indices with four positions (v3's render loop was ~100 VALU instructions per store: 1M x 11x11
    // 94-97 us, 77-79 with the strings; 512K x 18x18 109 -> 103-109).  v0 keeps the comparisons: the strings give the same
    // best time there (76.7-78.9 us) but only at narrow launch policies -- its extra VALU work paces the stores about right
    // ("a wait that staggers").  Two buffers, used by alternate chunks.
    constexpr bool MARKS = GT != 0 && V3;
    constexpr int GRPT = (GT & 1) ? 4 : 1, NPL = V3 ? 2 : 1;
    constexpr int WPG = GT ? (GRPT * GT * GT + 31) / 32 : 1;      // words per group string
    constexpr int NGW = GT ? (EPB / GRPT) * WPG : 1;              // words per plane
    uint32_t* marks = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(spawn) + ((CELLS * 2 + 15) & ~15));   // [2][NPL][NGW]
    __shared__ int spawn_count_s;

    const int tid = threadIdx.x;
    if (MARKS)
        for (int i = tid; i < 2 * NPL * NGW; i += LMAZE_BLOCK) marks[i] = 0u;
    // A workgroup takes chunks of EPB envs grid-stride (chunk = blockIdx.x, + gridDim.x, ...): with one chunk
    // each (gridDim.x = number of chunks) it is the plain one-chunk kernel; with several, the inputs of the
    // NEXT chunk are loaded while the current one is being stored, so their latency -- several microseconds
    // when they miss the caches and queue behind a saturated write stream -- is no longer exposed with only
    // 3 workgroups resident per CU.  Consecutive workgroups still write consecutive chunks at any moment.
    const int64_t nchunks = (a.n + EPB - 1) / EPB;
    int64_t chunk = blockIdx.x;
    int64_t blockbase = chunk * EPB;
    int nb = (int)min((int64_t)EPB, a.n - blockbase);  // envs of the current chunk
    const bool masked = !DO_STEP && a.mask != nullptr;
    const bool autoreset = DO_STEP && a.auto_reset;
    // the reset epoch, read HERE (a uniform load in front of every store: one scalar load): fetched where a done env
    // draws its placement it was a global round trip inside phase 1 for every wave with a done env -- 90.0 instead of
    // 83.4 us per step at 1M x 11x11 once the epoch lives on the device (captured rollouts)
    const uint64_t epoch = autoreset ? launch_epoch(a.epoch, a.epoch_in) : 0;
    if (autoreset) pass_epoch_on(a.epoch_in, a.epoch_out);

    EnvIn in{};
    if (tid < nb) in = load_env<VARIANT, DO_STEP>(a, blockbase + tid);
    // Streaming regime without the fused reset: the state loads are waited for BEFORE the set-up's layout loads go out.
    // That is a second round trip in series at the head of every workgroup, and it is faster: 1M x 11x11 at the default
    // policy 76.1-78.2 us on three boxes with it, 85-93 without (then only a narrow (5, 1) / (4, 2) reaches 77) --
    // the pause staggers the workgroups of a CU, fewer of them stream their ranges at the same moment (the regime
    // DESIGN 5.3 describes).  With the fused reset the set-up is longer and staggers by itself: 78.5-83 us without
    // the wait against 84-86 with it.  (Found when the done flag's `auto_reset ? byte : 0` select -- a use of a loaded
    // value inside the branch above -- turned out to have been that wait all along.)
    // It is a STAGGER, nothing else, and it is kept only while it pays: launch_hint bit 9 switches it off, and the guard
    // test times both (tests/test_gpu_policy_guard.py).
    if (DO_STEP && NT && !autoreset && !(a.launch_hint & 0x200)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // Large batches (NT): the first 256 workgroups touch every 64-byte line of this step's action row at kernel
    // start -- ONE burst of reads before the write stream saturates -- so that the chunk loads later in the
    // launch find the row in the memory-side cache.  Actions are the one input that is new every step (state
    // was written by the previous step and is still cached): read chunk by chunk from HBM they are 4 of 521
    // bytes per env but cost 17-25 % of the rate, each small read turning a saturated write stream around
    // (1M x 11x11, a fresh action row per step, 3 workgroups per CU x 2 chunks: 108 us without, 85-86 with;
    // 86 us when the rows are cache-resident anyway).
    // The same for the per-env state, which is only cached if nothing else ran since the previous step: with a
    // 512-MB copy between two steps (a stand-in for a policy network) the step took 131 us instead of 83.
    int warmed = 0;
    if (DO_STEP && NT) {
        warmed = warm_lines(a.action, a.n * 4, 256) + warm_lines(a.ball, a.n * 8, 256) +
                 warm_lines(a.step_count, a.n * 4, 256);
        if (!V3) warmed += warm_lines(a.reward, a.n * 4, 256);
        if (V3) warmed += warm_lines(a.goal, a.n * 8, 256);
        if (autoreset) warmed += warm_lines(a.done, a.n, 256);
    }

    constexpr int NSM = GT ? (GT * GT + 63) / 64 : 1;
    unsigned long long okm[NSM] = {};                             // GT != 0: accepted spawn cells, 64 per word
    int okcount = 0;
    if (GT != 0) {
        // Set-up with ONE global round trip: every layout byte this lane needs -- its cells for the LDS copy and the
        // pattern, and (wave 0, fused reset) the cells it ranks for the spawn list -- is loaded into registers before
        // the first is used.  Written as plain loops the compiler waits for each load where it is consumed: layout,
        // pattern, spawn list chunk 1, chunk 2 = four L2 round trips in series in front of the workgroup's first
        // barrier (two without the fused reset), about 0.5 us each with only 3 workgroups per CU to hide them.
        constexpr int CG = (GT ? GT : 1) * (GT ? GT : 1), NL = (CG + LMAZE_BLOCK - 1) / LMAZE_BLOCK, NS = (CG + 63) / 64;
        uint8_t cb[NL], sb[NS];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            cb[j] = a.layout[min(tid + j * LMAZE_BLOCK, CG - 1)];     // unconditional (clamped): a load under a branch is
        }                                                             // waited for where the branch ends
        if (autoreset) {                                              // uniform
#pragma unroll
            for (int j = 0; j < NS; ++j) sb[j] = a.layout[min((tid & 63) + j * 64, CG - 1)];
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int c = tid + j * LMAZE_BLOCK;
            if (c < CG) {
                lay[c] = cb[j];
                const int bits = cell_bits<VARIANT>(cb[j]);
                for (int g = 0; g < GRP; ++g) pat[g * CG + c] = bits;
            }
        }
        if (autoreset) {
            // the accepted spawn cells as NS ballots, bit c of the string = cell c accepted: every wave ranks the same
            // bytes, so the masks are wave-uniform scalars and nothing goes through LDS.  (Round 2: as a compacted
            // list built by wave 0 in front of the barrier the fused reset cost 4.7 of 83 us at 1M x 11x11 even
            // when no env was done -- the list is only read by the few lanes that reset.)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int c = (tid & 63) + j * 64;
                okm[j] = __ballot(c < CG && interior(c, G) && spawn_ok<VARIANT>(sb[j]));
                okcount += __popcll(okm[j]);
            }
        }
    } else {
        for (int i = tid; i < CELLS; i += LMAZE_BLOCK) lay[i] = a.layout[i];
        for (int i = tid; i < PAT; i += LMAZE_BLOCK) pat[i] = cell_bits<VARIANT>(a.layout[i]);
        if (autoreset && tid < 64) {
            const int cnt = wave_build_spawn_list<VARIANT>(a.layout, G, CELLS, spawn, tid);
            if (tid == 0) spawn_count_s = cnt;
        }
    }
    __syncthreads();

  int buf = 0;
  for (;;) {
    const int64_t next = chunk + gridDim.x;
    const bool has_next = next < nchunks;                         // uniform over the workgroup
    uint32_t* mk = marks + buf * NPL * NGW;
    EnvIn in_next{};
    int nb_next = 0;
    if (has_next) {                                               // issued now, consumed one chunk later
        nb_next = (int)min((int64_t)EPB, a.n - next * EPB);
        if (tid < nb_next) in_next = load_env<VARIANT, DO_STEP>(a, next * EPB + tid);
    }

    if (tid < EPB) {  // one lane per env
        const int le = tid;
        int bf = -8, gf = -8, mf = 0;
        if (le < nb) {
            const int64_t e = blockbase + le;
            int bc, gc;
            if (GT != 0) {
                env_phase1<VARIANT, DO_STEP>(a, lay, G, e, in, [&](uint4 d, int& pb, int& pg) { place_from_masks<VARIANT, NSM>(okm, okcount, d, pb, pg); }, epoch, bc, gc);
            } else {
                const int cnt = autoreset ? spawn_count_s : 0;
                env_phase1<VARIANT, DO_STEP>(a, lay, G, e, in, [&](uint4 d, int& pb, int& pg) { place_from_list<VARIANT>(spawn, cnt, d, pb, pg); }, epoch, bc, gc);
            }
            const int off = (GT != 0) ? (le % GRP) * CELLS : 0;
            bf = off + bc;
            if (V3 && gc >= 0) gf = off + gc;
            if (masked) mf = a.mask[e] != 0;
            if (MARKS) {
                const int gi = (le / GRP) * WPG;
                atomicOr(&mk[gi + (bf >> 5)], 1u << (bf & 31));
                if (V3 && gc >= 0) atomicOr(&mk[NGW + gi + (gf >> 5)], 1u << (gf & 31));
            }
        }
        ballflat[le] = bf;
        if (V3) goalflat[le] = gf;
        if (masked) maskflag[le] = mf;
    }
    if (a.obs != nullptr) {
    __syncthreads();

    int32_t* obs = a.obs + (size_t)blockbase * CELLS;
    int4* obs4 = reinterpret_cast<int4*>(obs);
    const int R = nb * CELLS;  // dwords this workgroup writes
    const int nq = R >> 2;     // whole 16-byte stores

    if (GT != 0) {
        const int V4G = PAT >> 2;  // stores per group of GRP envs
        const int4* pat4 = reinterpret_cast<const int4*>(pat);
        if (MARKS && has_next)                                    // the other buffer, for the next chunk's phase 1
            for (int i = tid; i < NPL * NGW; i += LMAZE_BLOCK) marks[(buf ^ 1) * NPL * NGW + i] = 0u;
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            const int grp = q / V4G;
            const int p = q - grp * V4G;
            int4 v = pat4[p];
            const int p4 = p << 2;
            if (masked) {
                if (GRP == 4) {
                    const int4 m = reinterpret_cast<const int4*>(maskflag)[grp];
                    if (!(m.x | m.y | m.z | m.w)) continue;
                } else if (!maskflag[grp]) {
                    continue;
                }
            }
            if (MARKS) {
                // dwords 4p .. 4p+3 of the group: one nibble of each string (4 divides 32: never across two words)
                const int w = grp * WPG + (p4 >> 5), sh = p4 & 31;
                const uint32_t b4 = mk[w] >> sh, g4 = mk[NGW + w] >> sh;
                v.x |= (int)(b4 & 1u) * LMAZE_OBS_BALL | (int)(g4 & 1u) * LMAZE_OBS_GOAL;
                v.y |= (int)((b4 >> 1) & 1u) * LMAZE_OBS_BALL | (int)((g4 >> 1) & 1u) * LMAZE_OBS_GOAL;
                v.z |= (int)((b4 >> 2) & 1u) * LMAZE_OBS_BALL | (int)((g4 >> 2) & 1u) * LMAZE_OBS_GOAL;
                v.w |= (int)((b4 >> 3) & 1u) * LMAZE_OBS_BALL | (int)((g4 >> 3) & 1u) * LMAZE_OBS_GOAL;
            } else if (GRP == 4) {
                const int4 bf = reinterpret_cast<const int4*>(ballflat)[grp];
                or_at(v, bf.x - p4, LMAZE_OBS_BALL);
                or_at(v, bf.y - p4, LMAZE_OBS_BALL);
                or_at(v, bf.z - p4, LMAZE_OBS_BALL);
                or_at(v, bf.w - p4, LMAZE_OBS_BALL);
            } else {
                or_at(v, ballflat[grp] - p4, LMAZE_OBS_BALL);
            }
            store16<NT>(obs4 + q, v);
        }
    } else {
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            const int f0 = q << 2;
            int le = f0 / CELLS;
            int c = f0 - le * CELLS;
            if (masked) {  // the store may straddle two envs
                const int le1 = (f0 + 3) / CELLS;
                if (!(maskflag[le] | maskflag[le1 < EPB ? le1 : le])) continue;
            }
            int vals[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = pat[c];
                v |= (ballflat[le] == c) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (goalflat[le] == c) ? LMAZE_OBS_GOAL : 0;
                vals[j] = v;
                if (++c == CELLS) { c = 0; ++le; }
            }
            store16<NT>(obs4 + q, make_int4(vals[0], vals[1], vals[2], vals[3]));
        }
    }
    // ragged tail (last workgroup only, when nb*G*G is not a multiple of 4)
    const int f = (nq << 2) + tid;
    if (f < R) {
        const int le = f / CELLS;
        const int c = f - le * CELLS;
        const int off = (GT != 0) ? (le % GRP) * CELLS : 0;
        int v = pat[(GT != 0 && GRP > 1) ? off + c : c];
        v |= (ballflat[le] == off + c) ? LMAZE_OBS_BALL : 0;
        if (V3) v |= (goalflat[le] == off + c) ? LMAZE_OBS_GOAL : 0;
        if (!masked || maskflag[le]) obs[f] = v;
    }
    }  // obs != nullptr
    if (!has_next) break;
    __syncthreads();   // the per-env cells in LDS are rewritten by the next chunk's phase 1
    chunk = next;
    blockbase = chunk * EPB;
    nb = nb_next;
    in = in_next;
    buf ^= 1;
  }
  if (warmed == 0x7fedcba9 && a.n < 0) a.done[0] = 1;   // never true: keeps the warming loads alive
}

// ------------------------------------------------------------------------------------
// Per-env layouts: uint8[N,G,G] in HBM, tiled into LDS per workgroup.
// ------------------------------------------------------------------------------------
template <int GT, int VARIANT, bool DO_STEP>
__global__ __launch_bounds__(LMAZE_BLOCK) void step_perenv_kernel(const StepArgs a) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    const int G = GT ? GT : a.grid;
    const int CELLS = G * G;
    const int EPB = a.envs_per_block;  // multiple of 16: keeps the tile base 16-byte aligned

    extern __shared__ int4 lds4[];
    uint8_t* tile = reinterpret_cast<uint8_t*>(lds4);  // [EPB*CELLS] layouts of this workgroup
    const int tile_bytes = (EPB * CELLS + 15) & ~15;
    int* ballcell = reinterpret_cast<int*>(tile + tile_bytes);  // [EPB+1]
    int* goalcell = ballcell + EPB + 1;                         // [EPB+1]
    int* flag = goalcell + EPB + 1;                             // [EPB+1] done-on-entry / observe mask
    int* newball = flag + EPB + 1;                              // [EPB]   auto-reset placement
    int* newgoal = newball + EPB;                               // [EPB]

    const int tid = threadIdx.x;
    const int64_t blockbase = (int64_t)blockIdx.x * EPB;
    const int nb = (int)min((int64_t)EPB, a.n - blockbase);
    const int R = nb * CELLS;  // layout bytes read == obs dwords written by this workgroup
    const bool autoreset = DO_STEP && a.auto_reset;
    const bool masked = !DO_STEP && a.mask != nullptr;
    // the reset epoch, read HERE (a uniform load in front of every store: one scalar load): fetched where a done env
    // draws its placement it was a global round trip inside phase 1 for every wave with a done env -- 90.0 instead of
    // 83.4 us per step at 1M x 11x11 once the epoch lives on the device (captured rollouts)
    const uint64_t epoch = autoreset ? launch_epoch(a.epoch, a.epoch_in) : 0;
    if (autoreset) pass_epoch_on(a.epoch_in, a.epoch_out);

    EnvIn in{};  // this lane's env, loaded while the layouts are still on their way
    if (tid < nb) in = load_env<VARIANT, DO_STEP>(a, blockbase + tid);
    if (tid <= EPB) {
        int f = 0;
        if (tid < nb) {
            if (autoreset) f = in.was_done != 0;
            if (masked) f = a.mask[blockbase + tid] != 0;
        }
        flag[tid] = f;
    }
    // stage the layouts: 16 B per lane per instruction, ragged byte tail
    const uint8_t* src = a.layout + (size_t)blockbase * CELLS;
    const int n16 = R >> 4;
    for (int i = tid; i < n16; i += LMAZE_BLOCK) lds4[i] = reinterpret_cast<const int4*>(src)[i];
    for (int i = (n16 << 4) + tid; i < R; i += LMAZE_BLOCK) tile[i] = src[i];
    __syncthreads();

    if (autoreset) {
        // wave w re-places the done envs w, w+4, ... of the workgroup with ballot scans of their tiles
        const int lane = tid & 63;
        for (int le = tid >> 6; le < nb; le += LMAZE_BLOCK / 64) {
            if (!flag[le]) continue;
            int bc, gc;
            wave_place<VARIANT>(tile + le * CELLS, G, CELLS, env_draw(a.seed, epoch, a.env_base + blockbase + le), lane,
                                bc, gc);
            if (lane == 0) { newball[le] = bc; newgoal[le] = gc; }
        }
        __syncthreads();
    }

    if (tid <= EPB) {
        int bcell = -8, gcell = -8;
        if (tid < nb) {
            const int64_t e = blockbase + tid;
            int2 b = in.b, g = in.g;
            if (DO_STEP) {
                int sc_in = in.sc;
                float r_in = in.r;
                if (autoreset && flag[tid]) {
                    const int bc = newball[tid], gc = newgoal[tid];
                    if (bc >= 0) b = make_int2(bc / G, bc % G);
                    if (V3 && gc >= 0) {
                        g = make_int2(gc / G, gc % G);
                        a.goal_rw[e] = g;
                    }
                    sc_in = 0;
                    r_in = -0.0f;
                }
                int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
                const int sc = sc_in + 1;
                int ox, oy;
                decode_action(in.act, ox, oy);
                const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
                float r;
                bool dn;
                if (transition_rule<VARIANT>(a, tile[tid * CELLS + tx * G + ty], ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by,
                                             r, dn) && a.goal_count)
                    count_goal(a.goal_count + e);
                a.ball[e] = make_int2(bx, by);
                a.step_count[e] = sc;
                a.reward[e] = r;
                a.done[e] = dn ? 1 : 0;
                b = make_int2(bx, by);
            } else {
                b = make_int2(clampi(b.x, 0, G - 1), clampi(b.y, 0, G - 1));
            }
            bcell = b.x * G + b.y;
            if (V3 && g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) gcell = g.x * G + g.y;
        }
        ballcell[tid] = bcell;
        if (V3) goalcell[tid] = gcell;
    }
    if (a.obs == nullptr) return;
    // large batches: non-temporal stores, as in the other kernels (1M x 11x11 per-env layouts, 8-KiB tiles: 121 -> 102 us)
    const bool stream = (size_t)a.n * CELLS * 4 > kNonTemporalObsBytes;   // uniform
    __syncthreads();

    int32_t* obs = a.obs + (size_t)blockbase * CELLS;
    int4* obs4 = reinterpret_cast<int4*>(obs);
    const int nq = R >> 2;
    const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
    const bool whole = (CELLS & 3) == 0;  // a 16-byte store never straddles two envs
    for (int q = tid; q < nq; q += LMAZE_BLOCK) {
        const uint32_t w = tile32[q];  // 4 layout cells
        const int f0 = q << 2;
        int le = f0 / CELLS;
        int c = f0 - le * CELLS;
        int vals[4];
        if (whole) {
            if (masked && !flag[le]) continue;
            const int bc = ballcell[le] - c;
            const int gc = V3 ? goalcell[le] - c : -8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = cell_bits<VARIANT>((uint8_t)(w >> (8 * j)));
                v |= (bc == j) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (gc == j) ? LMAZE_OBS_GOAL : 0;
                vals[j] = v;
            }
        } else {
            if (masked && !(flag[le] | flag[(f0 + 3) / CELLS])) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = cell_bits<VARIANT>((uint8_t)(w >> (8 * j)));
                v |= (ballcell[le] == c) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (goalcell[le] == c) ? LMAZE_OBS_GOAL : 0;
                vals[j] = v;
                if (++c == CELLS) { c = 0; ++le; }
            }
        }
        if (stream) store16<true>(obs4 + q, make_int4(vals[0], vals[1], vals[2], vals[3]));
        else obs4[q] = make_int4(vals[0], vals[1], vals[2], vals[3]);
    }
    const int f = (nq << 2) + tid;
    if (f < R) {
        const int le = f / CELLS;
        const int c = f - le * CELLS;
        int v = cell_bits<VARIANT>(tile[f]);
        v |= (ballcell[le] == c) ? LMAZE_OBS_BALL : 0;
        if (V3) v |= (goalcell[le] == c) ? LMAZE_OBS_GOAL : 0;
        if (!masked || flag[le]) obs[f] = v;
    }
}

// ------------------------------------------------------------------------------------
// Per-env layouts, register-tiled: one WAVE per env, no LDS, no barrier (G*G a multiple of
// 256: G = 16, 32, 48, 64).  Lane l loads dword 64*j + l of the env's layout (256 B per wave
// instruction, each byte read from HBM once) and keeps it in registers; the target cell of
// the transition is fetched with v_readlane; every quantity of the transition is
// wave-uniform, lane 0 stores it; the lane's 4 cells per register become one 16-byte store,
// so every wave store instruction writes 1 KiB contiguous.  Waves never wait for each other.
// ------------------------------------------------------------------------------------
template <int G, int VARIANT, bool DO_STEP, bool NT>
__global__ __launch_bounds__(LMAZE_BLOCK) void step_perenv_wave_kernel(const StepArgs a) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    constexpr int CELLS = G * G, NJ = CELLS / 256;
    static_assert(CELLS % 256 == 0, "whole dwords per lane");
    const int lane = threadIdx.x & 63;
    const int EPW = a.envs_per_block;  // envs per wave
    const int64_t wave = (int64_t)blockIdx.x * (LMAZE_BLOCK / 64) + (threadIdx.x >> 6);
    const bool autoreset = DO_STEP && a.auto_reset;
    const bool masked = !DO_STEP && a.mask != nullptr;
    // the reset epoch, read HERE (a uniform load in front of every store: one scalar load): fetched where a done env
    // draws its placement it was a global round trip inside phase 1 for every wave with a done env -- 90.0 instead of
    // 83.4 us per step at 1M x 11x11 once the epoch lives on the device (captured rollouts)
    const uint64_t epoch = autoreset ? launch_epoch(a.epoch, a.epoch_in) : 0;
    if (autoreset) pass_epoch_on(a.epoch_in, a.epoch_out);

#pragma unroll 1
    for (int k = 0; k < EPW; ++k) {
        const int64_t e = wave * EPW + k;
        if (e >= a.n) break;
        if (masked && !a.mask[e]) continue;
        const uint32_t* lay32 = reinterpret_cast<const uint32_t*>(a.layout + (size_t)e * CELLS);
        uint32_t w[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = lay32[j * 64 + lane];

        int2 b = a.ball[e];
        int2 g = make_int2(-1, -1);
        if (V3) g = a.goal[e];
        if (DO_STEP) {
            const int act = a.action[e];
            int sc_in = a.step_count[e];
            float r_in = V3 ? 0.0f : a.reward[e];
            if (autoreset && a.done[e]) {  // re-place from the layout registers (wave-uniform branch)
                int bc, gc;
                wave_place_regs<VARIANT, G, NJ>(w, env_draw(a.seed, epoch, a.env_base + e), lane, bc, gc);
                if (bc >= 0) b = make_int2(bc / G, bc % G);
                if (V3 && gc >= 0) {
                    g = make_int2(gc / G, gc % G);
                    if (lane == 0) a.goal_rw[e] = g;
                }
                sc_in = 0;
                r_in = -0.0f;
            }
            int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
            int ox, oy;
            decode_action(act, ox, oy);
            const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
            const int t = __builtin_amdgcn_readfirstlane(tx * G + ty);  // wave-uniform by construction
            const int tw = t >> 2;                                      // dword of the layout holding the target cell
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if ((tw >> 6) == j) word = __builtin_amdgcn_readlane(w[j], tw & 63);
            const uint8_t c = (uint8_t)(word >> (8 * (t & 3)));
            float r;
            bool dn;
            const int sc = sc_in + 1;
            const bool hit = transition_rule<VARIANT>(a, c, ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn);
            if (lane == 0) {
                if (hit && a.goal_count) count_goal(a.goal_count + e);
                a.ball[e] = make_int2(bx, by);
                a.step_count[e] = sc;
                a.reward[e] = r;
                a.done[e] = dn ? 1 : 0;
            }
            b = make_int2(bx, by);
        } else {
            b = make_int2(clampi(b.x, 0, G - 1), clampi(b.y, 0, G - 1));
        }
        if (a.obs == nullptr) continue;
        const int bcell = b.x * G + b.y;
        const int gcell = (V3 && g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
        int4* obs4 = reinterpret_cast<int4*>(a.obs + (size_t)e * CELLS);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c0 = (j * 64 + lane) << 2;  // first of this lane's 4 cells
            int vals[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int v = cell_bits<VARIANT>((uint8_t)(w[j] >> (8 * q)));
                v |= (bcell == c0 + q) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (gcell == c0 + q) ? LMAZE_OBS_GOAL : 0;
                vals[q] = v;
            }
            store16<NT, 2>(obs4 + j * 64 + lane, make_int4(vals[0], vals[1], vals[2], vals[3]));   // `sc0 sc1 nt`: -5.6 % here (lmaze_common.h)
        }
    }
}

// ------------------------------------------------------------------------------------
// Shared 8x8 layout, small batches (BASELINE config 2: 65 536 x 8x8, 16.8 MB of planes): wave-autonomous.
// At this size a launch is nothing but a latency chain -- 5.7-6.0 us per step against 4.6 us for a bare fill
// of the same bytes -- so the chain is cut to its minimum: no LDS, no workgroup barrier.  One wave = 64 envs;
// lane l holds the layout byte of cell l (the 64 cells ARE the wave), so the transition's target cell and the
// render's pattern come from other lanes by ds_bpermute; the wave's 64 x 256 B of planes are one contiguous
// 16 KiB, 16 stores per lane, store k of lane l covering cells 4(l % 16)..+3 of env l/16 + 4k -- the pattern
// int4 is loop-invariant per lane.  Fused auto-reset: the accepted spawn cells are a 64-bit ballot.
// ------------------------------------------------------------------------------------

// EPW = envs per wave (64, 32 or 16: fewer envs per wave = more, shorter waves to hide the load latency with)
// Tried and dropped (round 2): storing the ball-free planes first -- they need the 64-byte layout only -- and
// patching the ball cell's dword once the inputs have arrived, to hide the load latency behind the stores: 6.5-7.7 us
// instead of 5.6 (one partial-line store per env costs more than the latency it hides).
template <int VARIANT, bool DO_STEP, int EPW>
__global__ __launch_bounds__(LMAZE_BLOCK) void step_shared_wave8_kernel(const StepArgs a) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    constexpr int G = 8, CELLS = 64;
    const int lane = threadIdx.x & 63;
    // blockDim.x / 64 autonomous waves per workgroup (no barrier, no LDS: the grouping only changes what the dispatcher sees)
    const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * EPW;
    if (base >= a.n) return;
    const int nb = (int)min((int64_t)EPW, a.n - base);
    const bool autoreset = DO_STEP && a.auto_reset;
    // the reset epoch, read HERE (a uniform load in front of every store: one scalar load): fetched where a done env
    // draws its placement it was a global round trip inside phase 1 for every wave with a done env -- 90.0 instead of
    // 83.4 us per step at 1M x 11x11 once the epoch lives on the device (captured rollouts)
    const uint64_t epoch = autoreset ? launch_epoch(a.epoch, a.epoch_in) : 0;
    if (autoreset) pass_epoch_on(a.epoch_in, a.epoch_out);
    const bool live = lane < nb;
    const int myc = a.layout[lane];                                   // cell `lane` of the layout; issued first
    EnvIn in{};
    in.b = make_int2(1, 1); in.g = make_int2(-1, -1);
    if (live) in = load_env<VARIANT, DO_STEP>(a, base + lane);
    const int mypat = cell_bits<VARIANT>((uint8_t)myc);
    const int p4 = (lane & 15) << 2;                                  // first cell of this lane's stores
    const int4 pat4 = make_int4(__shfl(mypat, p4, 64), __shfl(mypat, p4 + 1, 64), __shfl(mypat, p4 + 2, 64),
                                __shfl(mypat, p4 + 3, 64));
    int4* obs4 = reinterpret_cast<int4*>(a.obs + (size_t)base * CELLS);
    int2 b = in.b, g = in.g;
    if (DO_STEP) {
        int sc_in = in.sc;
        float r_in = in.r;
        if (autoreset) {   // reference reset(): placement over the accepted cells, ranked row-major (lmaze_common.h place_from_list)
            const unsigned long long ok = __ballot(interior(lane, G) && spawn_ok<VARIANT>((uint8_t)myc));
            if (in.was_done) {
                const uint4 d = env_draw(a.seed, epoch, a.env_base + base + lane);
                const int count = __popcll(ok);
                if (V3) {
                    int kg = -1;
                    if (count > 0) {
                        kg = (int)__umulhi(d.x, (uint32_t)count);
                        const int gc = kth_set_bit(ok, kg);
                        g = make_int2(gc / G, gc % G);
                        if (live) a.goal_rw[base + lane] = g;
                    }
                    if (count > 1) {
                        int kb = (int)__umulhi(d.y, (uint32_t)(count - 1));
                        kb += (kb >= kg);
                        const int bc = kth_set_bit(ok, kb);
                        b = make_int2(bc / G, bc % G);
                    }
                } else if (count > 0) {
                    const int bc = kth_set_bit(ok, (int)__umulhi(d.y, (uint32_t)count));
                    b = make_int2(bc / G, bc % G);
                }
                sc_in = 0;      // v0:110
                r_in = -0.0f;   // v0:109
            }
        }
        int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
        const int sc = sc_in + 1;  // v0:151, v3:225
        int ox, oy;
        decode_action(in.act, ox, oy);
        const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
        const uint8_t c = (uint8_t)__shfl(myc, tx * G + ty, 64);      // v0:172, v3:251 -- every lane takes part
        float r;
        bool dn;
        const bool hit = transition_rule<VARIANT>(a, c, ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn);
        if (live) {
            const int64_t e = base + lane;
            if (hit && a.goal_count) count_goal(a.goal_count + e);
            a.ball[e] = make_int2(bx, by);
            a.step_count[e] = sc;
            a.reward[e] = r;
            a.done[e] = dn ? 1 : 0;
        }
        b = make_int2(bx, by);
    } else {
        b = make_int2(clampi(b.x, 0, G - 1), clampi(b.y, 0, G - 1));
    }
    if (a.obs == nullptr) return;
    const int ball_cell = b.x * G + b.y;
    const int goal_cell = (V3 && g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
#pragma unroll
    for (int k = 0; k < EPW / 4; ++k) {
        const int le = (lane >> 4) + 4 * k;                           // env of store k
        const int bc = __shfl(ball_cell, le, 64);
        const int gc = V3 ? __shfl(goal_cell, le, 64) : -8;
        int4 v = pat4;
        or_at(v, bc - p4, LMAZE_OBS_BALL);
        if (V3) or_at(v, gc - p4, LMAZE_OBS_GOAL);
        if (le < nb) obs4[lane + 64 * k] = v;
    }
}

// ------------------------------------------------------------------------------------
// Narrow observation: the same bit mask in ONE BYTE per cell, uint8[N,G,G] -- the mask needs four
// bits --, 37 + G*G bytes per env-step instead of 37 + 4 G*G (11x11: 158 against 521).  The int32 planes stay the
// metric's mode; this is a separate workload with its own algorithmic bytes.  Shared layout, any G.
// Phase 1 is the int32 kernel's (env_phase1); the render stripes the workgroup's EPB * G*G bytes with 16-byte stores,
// 16 cells each, which straddle envs at every odd G: the static pattern comes from four byte-shifted copies of the
// layout's bits laid out twice in a row (so that any 16 consecutive cells, wrapping into the next env, are four aligned
// dword reads), the ball / goal bytes of the one or two envs a store touches are OR-ed in.
// ------------------------------------------------------------------------------------
template <int VARIANT, bool DO_STEP, int EPB>
__global__ __launch_bounds__(LMAZE_BLOCK) void step_shared_u8_kernel(const StepArgs a) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    static_assert(EPB % 16 == 0 && EPB <= LMAZE_BLOCK, "a workgroup's byte range starts on a 16-byte boundary; one lane per env");
    const int G = a.grid, CELLS = G * G;
    const int PW = (2 * CELLS + 16 + 3) >> 2;             // dwords of one shifted copy: the pattern twice + 16 bytes of slack
    extern __shared__ int4 lds4[];
    uint32_t* patw = reinterpret_cast<uint32_t*>(lds4);                 // [4][PW] copy s = the doubled pattern starting at byte s
    int* ballflat = reinterpret_cast<int*>(patw + 4 * PW);              // [EPB + 1]
    int* goalflat = ballflat + EPB + 1;                                 // [EPB + 1]
    int* maskflag = goalflat + EPB + 1;                                 // [EPB + 1]
    uint16_t* spawn = reinterpret_cast<uint16_t*>(maskflag + EPB + 1);  // [CELLS]
    uint8_t* lay = reinterpret_cast<uint8_t*>(spawn + ((CELLS + 1) & ~1));   // [CELLS]
    __shared__ int spawn_count_s;
    const int tid = threadIdx.x;
    // chunks of EPB envs grid-stride, as step_shared_kernel: the set-up is paid once per workgroup and the next chunk's
    // per-env inputs are loaded while the current one is rendered
    const int64_t nchunks = (a.n + EPB - 1) / EPB;
    int64_t chunk = blockIdx.x;
    int64_t blockbase = chunk * EPB;
    int nb = (int)min((int64_t)EPB, a.n - blockbase);
    const bool masked = !DO_STEP && a.mask != nullptr;
    const bool autoreset = DO_STEP && a.auto_reset;
    const uint64_t epoch = autoreset ? launch_epoch(a.epoch, a.epoch_in) : 0;
    if (autoreset) pass_epoch_on(a.epoch_in, a.epoch_out);
    EnvIn in{};
    if (tid < nb) in = load_env<VARIANT, DO_STEP>(a, blockbase + tid);
    // large batches: the first 256 workgroups touch every line of this step's per-env inputs once, as step_shared_kernel does
    int warmed = 0;
    if (DO_STEP && (a.launch_hint & 0x100)) {
        warmed = warm_lines(a.action, a.n * 4, 256) + warm_lines(a.ball, a.n * 8, 256) + warm_lines(a.step_count, a.n * 4, 256);
        if (!V3) warmed += warm_lines(a.reward, a.n * 4, 256);
        if (V3) warmed += warm_lines(a.goal, a.n * 8, 256);
        if (autoreset) warmed += warm_lines(a.done, a.n, 256);
    }
    // set-up: ONE global round trip (the layout bytes, beside the env loads above), then everything from LDS
    for (int i = tid; i < CELLS; i += LMAZE_BLOCK) lay[i] = a.layout[i];
    __syncthreads();
    uint8_t* patb = reinterpret_cast<uint8_t*>(patw);
    for (int i = tid; i < 4 * PW; i += LMAZE_BLOCK) {                   // one dword of one copy per lane-iteration
        const int s = i / PW, k = (i - s * PW) << 2;                    // bytes k .. k+3 of copy s = pattern bytes k+s .. of the doubled row
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = k + s + j;
            c -= c >= CELLS ? CELLS : 0;
            c -= c >= CELLS ? CELLS : 0;
            c = min(c, CELLS - 1);                                       // slack bytes past the doubled row: never selected
            w |= (uint32_t)cell_bits<VARIANT>(lay[c]) << (8 * j);
        }
        patw[i] = w;
    }
    if (autoreset && tid < 64) {
        const int cnt = wave_build_spawn_list<VARIANT>(lay, G, CELLS, spawn, tid);
        if (tid == 0) spawn_count_s = cnt;
    }
    if (tid == 0) { ballflat[EPB] = -64; goalflat[EPB] = -64; maskflag[EPB] = 0; }
    __syncthreads();
  for (;;) {
    const int64_t next = chunk + gridDim.x;
    const bool has_next = next < nchunks;                               // uniform over the workgroup
    EnvIn in_next{};
    int nb_next = 0;
    if (has_next) {
        nb_next = (int)min((int64_t)EPB, a.n - next * EPB);
        if (tid < nb_next) in_next = load_env<VARIANT, DO_STEP>(a, next * EPB + tid);
    }
    if (tid < EPB) {
        int bf = -64, gf = -64, mf = 0;
        if (tid < nb) {
            const int64_t e = blockbase + tid;
            int bc, gc;
            const int cnt = autoreset ? spawn_count_s : 0;
            env_phase1<VARIANT, DO_STEP>(a, lay, G, e, in, [&](uint4 d, int& pb, int& pg) { place_from_list<VARIANT>(spawn, cnt, d, pb, pg); }, epoch, bc, gc);
            bf = bc;
            if (V3 && gc >= 0) gf = gc;
            if (masked) mf = a.mask[e] != 0;
        }
        ballflat[tid] = bf;
        goalflat[tid] = gf;
        maskflag[tid] = mf;
    }
    if (a.obs8 != nullptr) {
    __syncthreads();
    uint8_t* obs = a.obs8 + (size_t)blockbase * CELLS;
    const int R = nb * CELLS;
    if (!masked) {
        const int nq = R >> 4;
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            const int f0 = q << 4;
            const int le = f0 / CELLS, c = f0 - le * CELLS;             // first cell of the store
            const int sft = c & 3;
            const uint32_t* src = patw + sft * PW + ((c - sft) >> 2);
            uint32_t w[4] = {src[0], src[1], src[2], src[3]};
            // ball (and v3 goal) bytes of env le at byte b - c, of env le + 1 at byte CELLS - c + b'
            const int b0 = ballflat[le] - c, b1 = CELLS - c + ballflat[min(le + 1, EPB)];
            if ((unsigned)b0 < 16u) w[b0 >> 2] |= (uint32_t)LMAZE_OBS_BALL << ((b0 & 3) << 3);
            if ((unsigned)b1 < 16u && le + 1 < nb) w[b1 >> 2] |= (uint32_t)LMAZE_OBS_BALL << ((b1 & 3) << 3);
            if (V3) {
                const int g0 = goalflat[le] - c, g1 = CELLS - c + goalflat[min(le + 1, EPB)];
                if ((unsigned)g0 < 16u) w[g0 >> 2] |= (uint32_t)LMAZE_OBS_GOAL << ((g0 & 3) << 3);
                if ((unsigned)g1 < 16u && le + 1 < nb) w[g1 >> 2] |= (uint32_t)LMAZE_OBS_GOAL << ((g1 & 3) << 3);
            }
            if (a.launch_hint & 0x100) {                                // set by the launcher for large batches: non-temporal stores
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u t = {w[0], w[1], w[2], w[3]};
                stream_store16(reinterpret_cast<v4u*>(obs) + q, t);
            } else {
                reinterpret_cast<uint4*>(obs)[q] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        for (int f = (nq << 4) + tid; f < R; f += LMAZE_BLOCK) {        // ragged tail of the last chunk
            const int le = f / CELLS, c = f - le * CELLS;
            obs[f] = (uint8_t)(patb[c] | (ballflat[le] == c ? LMAZE_OBS_BALL : 0) | (V3 && goalflat[le] == c ? LMAZE_OBS_GOAL : 0));
        }
    } else {
        for (int f = tid; f < R; f += LMAZE_BLOCK) {                    // masked re-render (after a masked reset): byte stores
            const int le = f / CELLS, c = f - le * CELLS;
            if (!maskflag[le]) continue;
            obs[f] = (uint8_t)(patb[c] | (ballflat[le] == c ? LMAZE_OBS_BALL : 0) | (V3 && goalflat[le] == c ? LMAZE_OBS_GOAL : 0));
        }
    }
    }
    if (!has_next) break;
    __syncthreads();                                                     // the per-env cells in LDS are rewritten by the next chunk
    chunk = next;
    blockbase = chunk * EPB;
    nb = nb_next;
    in = in_next;
  }
    if (warmed == 0x7fedcba9 && a.n < 0) a.done[0] = 1;   // never true: keeps the warming loads alive
}

// ------------------------------------------------------------------------------------
// T steps in ONE launch for the same on-die 8x8 batches (lmaze_rollout_v0 / _v3): the wave-autonomous kernel above with
// the step loop inside.  A wave's 64 envs stay in registers for the whole rollout -- envs are independent, nothing
// synchronises --, each step's action row is fetched one step ahead, the planes are rewritten every step exactly as T
// launches would (same 16 KiB per wave: they stay in L2) and the per-env state goes back once at the end; optional
// per-step reward / done rows for the caller who needs the trajectory.  Step t draws its placements with epoch + t, as
// the t-th of T lmaze_step_*_autoreset calls would.  Bit-identical to those T calls.
// ------------------------------------------------------------------------------------
struct RolloutArgs {
    const int32_t* actions;   // [T, N]
    float* reward_t;          // [T, N] or null
    uint8_t* done_t;          // [T, N] or null
    int32_t T;
};

template <int VARIANT, int EPW>
__global__ __launch_bounds__(LMAZE_BLOCK) void rollout_shared_wave8_kernel(const StepArgs a, const RolloutArgs ro) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    constexpr int G = 8, CELLS = 64;
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * EPW;
    if (base >= a.n) return;
    const int nb = (int)min((int64_t)EPW, a.n - base);
    const bool autoreset = a.auto_reset != 0;
    const bool live = lane < nb;
    const int64_t e = base + lane;
    const int myc = a.layout[lane];
    int2 b = make_int2(1, 1), g = make_int2(-1, -1);
    int sc = 0, was_done = 0, hits = 0, act_next = -1;
    float r = 0.0f;
    if (live) {
        b = a.ball[e];
        if (V3) g = a.goal[e];
        sc = a.step_count[e];
        if (!V3) r = a.reward[e];
        if (autoreset) was_done = a.done[e];
        act_next = ro.actions[e];
    }
    const int mypat = cell_bits<VARIANT>((uint8_t)myc);
    const int p4 = (lane & 15) << 2;
    const int4 pat4 = make_int4(__shfl(mypat, p4, 64), __shfl(mypat, p4 + 1, 64), __shfl(mypat, p4 + 2, 64),
                                __shfl(mypat, p4 + 3, 64));
    const unsigned long long ok = __ballot(interior(lane, G) && spawn_ok<VARIANT>((uint8_t)myc));
    int4* obs4 = a.obs ? reinterpret_cast<int4*>(a.obs + (size_t)base * CELLS) : nullptr;
    bool dn = false;
    for (int t = 0; t < ro.T; ++t) {
        const int act = act_next;
        if (t + 1 < ro.T && live) act_next = ro.actions[(size_t)(t + 1) * a.n + e];      // next step's row, in flight over this step
        float r_in = r;
        if (autoreset && was_done) {   // reference reset(): as step_shared_wave8_kernel, with this step's epoch
            const uint4 d = env_draw(a.seed, a.epoch + (uint64_t)t, a.env_base + e);
            const int count = __popcll(ok);
            if (V3) {
                int kg = -1;
                if (count > 0) {
                    kg = (int)__umulhi(d.x, (uint32_t)count);
                    const int gc = kth_set_bit(ok, kg);
                    g = make_int2(gc / G, gc % G);
                }
                if (count > 1) {
                    int kb = (int)__umulhi(d.y, (uint32_t)(count - 1));
                    kb += (kb >= kg);
                    const int bc = kth_set_bit(ok, kb);
                    b = make_int2(bc / G, bc % G);
                }
            } else if (count > 0) {
                const int bc = kth_set_bit(ok, (int)__umulhi(d.y, (uint32_t)count));
                b = make_int2(bc / G, bc % G);
            }
            sc = 0;         // v0:110
            r_in = -0.0f;   // v0:109
        }
        int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
        sc += 1;            // v0:151, v3:225
        int ox, oy;
        decode_action(act, ox, oy);
        const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
        const uint8_t c = (uint8_t)__shfl(myc, tx * G + ty, 64);      // v0:172, v3:251 -- every lane takes part
        const bool hit = transition_rule<VARIANT>(a, c, ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn);
        hits += hit ? 1 : 0;
        b = make_int2(bx, by);
        was_done = dn ? 1 : 0;
        if (live) {
            if (ro.reward_t) ro.reward_t[(size_t)t * a.n + e] = r;
            if (ro.done_t) ro.done_t[(size_t)t * a.n + e] = dn ? 1 : 0;
        }
        if (obs4) {
            const int ball_cell = b.x * G + b.y;
            const int goal_cell = (V3 && g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
#pragma unroll
            for (int k = 0; k < EPW / 4; ++k) {
                const int le = (lane >> 4) + 4 * k;
                const int bc = __shfl(ball_cell, le, 64);
                const int gc = V3 ? __shfl(goal_cell, le, 64) : -8;
                int4 v = pat4;
                or_at(v, bc - p4, LMAZE_OBS_BALL);
                if (V3) or_at(v, gc - p4, LMAZE_OBS_GOAL);
                if (le < nb) obs4[lane + 64 * k] = v;
            }
        }
    }
    if (live && ro.T > 0) {
        a.ball[e] = b;
        if (V3 && autoreset) a.goal_rw[e] = g;
        a.step_count[e] = sc;
        a.reward[e] = r;
        a.done[e] = dn ? 1 : 0;
        if (hits && a.goal_count) a.goal_count[e] += hits;
    }
}

// T steps of a shared-layout batch of ANY grid size in one launch, for batches whose planes stay on-die: a workgroup owns
// EPB envs for the whole rollout -- one lane per env keeps ball, goal, stepCount, reward and done in registers across the
// T steps, the layout, its plane pattern and the spawn list are set up in LDS once -- and re-renders its envs' planes after
// every step exactly as T launches of step_shared_kernel would (two barriers per step, no grid-wide one: envs are
// independent).  Per-step launches of such a batch are launch-bound (65 536 x 11x11: 8 us per launch for 32 MB of planes
// that never leave the caches).  Transition, fused reset (placement from the compacted spawn list with the draw of
// epoch + t) and the v0 reward persistence are the step kernel's own functions: bit-identical by construction and by test.
template <int VARIANT>
__global__ __launch_bounds__(LMAZE_BLOCK) void rollout_shared_kernel(const StepArgs a, const RolloutArgs ro) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    const int G = a.grid, CELLS = G * G, EPB = a.envs_per_block;
    extern __shared__ int4 lds4[];
    int* pat = reinterpret_cast<int*>(lds4);                                  // [CELLS] plane bits without ball / goal
    int* ballflat = pat + CELLS;                                              // [EPB]
    int* goalflat = ballflat + EPB;                                           // [EPB]
    uint8_t* lay = reinterpret_cast<uint8_t*>(goalflat + EPB);                // [CELLS]
    uint16_t* spawn = reinterpret_cast<uint16_t*>(lay + ((CELLS + 15) & ~15));   // [CELLS] accepted spawn cells, row-major
    __shared__ int spawn_count_s;

    const int tid = threadIdx.x;
    const int64_t blockbase = (int64_t)blockIdx.x * EPB;
    const int nb = (int)min((int64_t)EPB, a.n - blockbase);
    const bool autoreset = a.auto_reset != 0, live = tid < nb;
    const int64_t e = blockbase + tid;
    int2 b = make_int2(1, 1), g = make_int2(-1, -1);
    int sc = 0, was_done = 0, hits = 0, act_next = -1;
    float r = 0.0f;
    if (live) {                                                               // in flight over the set-up
        b = a.ball[e];
        if (V3) g = a.goal[e];
        sc = a.step_count[e];
        if (!V3) r = a.reward[e];
        if (autoreset) was_done = a.done[e];
        act_next = ro.actions[e];
    }
    for (int i = tid; i < CELLS; i += LMAZE_BLOCK) {
        const uint8_t c = a.layout[i];
        lay[i] = c;
        pat[i] = cell_bits<VARIANT>(c);
    }
    if (autoreset && tid < 64) {
        const int cnt = wave_build_spawn_list<VARIANT>(a.layout, G, CELLS, spawn, tid);
        if (tid == 0) spawn_count_s = cnt;
    }
    __syncthreads();
    const int spawn_count = autoreset ? spawn_count_s : 0;

    int32_t* obs = a.obs ? a.obs + (size_t)blockbase * CELLS : nullptr;
    const int R = nb * CELLS, nq = R >> 2;
    bool dn = false;
    for (int t = 0; t < ro.T; ++t) {
        if (live) {
            const int act = act_next;
            if (t + 1 < ro.T) act_next = ro.actions[(size_t)(t + 1) * a.n + e];       // next step's row, in flight over this step
            float r_in = r;
            if (autoreset && was_done) {                                              // reference reset(), as env_phase1
                int bc, gc;
                place_from_list<VARIANT>(spawn, spawn_count, env_draw(a.seed, a.epoch + (uint64_t)t, a.env_base + e), bc, gc);
                if (bc >= 0) b = make_int2(bc / G, bc % G);
                if (V3 && gc >= 0) g = make_int2(gc / G, gc % G);
                sc = 0;         // v0:110
                r_in = -0.0f;   // v0:109
            }
            int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
            sc += 1;            // v0:151, v3:225
            int ox, oy;
            decode_action(act, ox, oy);
            const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
            hits += transition_rule<VARIANT>(a, lay[tx * G + ty], ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn) ? 1 : 0;
            b = make_int2(bx, by);
            was_done = dn ? 1 : 0;
            if (ro.reward_t) ro.reward_t[(size_t)t * a.n + e] = r;
            if (ro.done_t) ro.done_t[(size_t)t * a.n + e] = dn ? 1 : 0;
            ballflat[tid] = b.x * G + b.y;
            if (V3) goalflat[tid] = (g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
        }
        if (obs == nullptr) continue;                                                 // uniform
        __syncthreads();
        // dword f of the workgroup's range = cell f % CELLS of env f / CELLS; a lane walks its stores 1024 dwords apart
        int le = (tid << 2) / CELLS, c = (tid << 2) - le * CELLS;
        const int dle = (LMAZE_BLOCK << 2) / CELLS, dc = (LMAZE_BLOCK << 2) - dle * CELLS;
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            int vals[4], l2 = le, c2 = c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = pat[c2];
                v |= (ballflat[l2] == c2) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (goalflat[l2] == c2) ? LMAZE_OBS_GOAL : 0;
                vals[j] = v;
                if (++c2 == CELLS) { c2 = 0; ++l2; }
            }
            reinterpret_cast<int4*>(obs)[q] = make_int4(vals[0], vals[1], vals[2], vals[3]);
            le += dle; c += dc;
            if (c >= CELLS) { c -= CELLS; ++le; }
        }
        const int f = (nq << 2) + tid;                                                // ragged tail: nb*G*G not a multiple of 4
        if (f < R) {
            const int l2 = f / CELLS, c2 = f - l2 * CELLS;
            int v = pat[c2];
            v |= (ballflat[l2] == c2) ? LMAZE_OBS_BALL : 0;
            if (V3) v |= (goalflat[l2] == c2) ? LMAZE_OBS_GOAL : 0;
            obs[f] = v;
        }
        __syncthreads();                                                              // ballflat / goalflat are rewritten by the next step
    }
    if (live && ro.T > 0) {
        a.ball[e] = b;
        if (V3 && autoreset) a.goal_rw[e] = g;
        a.step_count[e] = sc;
        a.reward[e] = r;
        a.done[e] = dn ? 1 : 0;
        if (hits && a.goal_count) a.goal_count[e] += hits;
    }
}

// The same for per-env layouts: the workgroup's EPB layouts (EPB * G * G bytes) are copied to LDS once and serve the
// collision check, the planes and -- a done env, fused reset -- the whole-wave placement on the env's own maze (wave_place,
// the rule of the per-env step kernels) for all T steps.  EPB <= 64: every env's lane sits in wave 0, which places the
// done envs one after the other.
template <int VARIANT>
__global__ __launch_bounds__(LMAZE_BLOCK) void rollout_perenv_kernel(const StepArgs a, const RolloutArgs ro) {
    constexpr bool V3 = VARIANT == LMAZE_VARIANT_V3;
    const int G = a.grid, CELLS = G * G, EPB = a.envs_per_block;
    extern __shared__ int4 lds4[];
    int* ballflat = reinterpret_cast<int*>(lds4);                             // [EPB] cell of the ball
    int* goalflat = ballflat + EPB;                                           // [EPB]
    uint8_t* lays = reinterpret_cast<uint8_t*>(goalflat + EPB);               // [EPB * CELLS]

    const int tid = threadIdx.x;
    const int64_t blockbase = (int64_t)blockIdx.x * EPB;
    const int nb = (int)min((int64_t)EPB, a.n - blockbase);
    const bool autoreset = a.auto_reset != 0, live = tid < nb;
    const int64_t e = blockbase + tid;
    int2 b = make_int2(1, 1), g = make_int2(-1, -1);
    int sc = 0, was_done = 0, hits = 0, act_next = -1;
    float r = 0.0f;
    if (live) {
        b = a.ball[e];
        if (V3) g = a.goal[e];
        sc = a.step_count[e];
        if (!V3) r = a.reward[e];
        if (autoreset) was_done = a.done[e];
        act_next = ro.actions[e];
    }
    {   // EPB is a multiple of 4: the workgroup's layouts start on a dword and are whole dwords
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.layout + (size_t)blockbase * CELLS);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lays);
        const int nw = (nb * CELLS) >> 2;
        for (int i = tid; i < nw; i += LMAZE_BLOCK) dst[i] = src[i];
        for (int i = (nw << 2) + tid; i < nb * CELLS; i += LMAZE_BLOCK) lays[i] = a.layout[(size_t)blockbase * CELLS + i];
    }
    __syncthreads();

    int32_t* obs = a.obs ? a.obs + (size_t)blockbase * CELLS : nullptr;
    const int R = nb * CELLS, nq = R >> 2;
    bool dn = false;
    for (int t = 0; t < ro.T; ++t) {
        if (tid < 64) {                                                               // wave 0, every lane: the ballots below
            const int act = act_next;
            if (live && t + 1 < ro.T) act_next = ro.actions[(size_t)(t + 1) * a.n + e];
            float r_in = r;
            // reference reset() of the done envs, one whole-wave placement each on the env's own layout
            unsigned long long todo = __ballot(live && autoreset && was_done);
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                int bc, gc;
                wave_place<VARIANT>(lays + j * CELLS, G, CELLS, env_draw(a.seed, a.epoch + (uint64_t)t, a.env_base + blockbase + j), tid, bc, gc);
                if (tid == j) {
                    if (bc >= 0) b = make_int2(bc / G, bc % G);
                    if (V3 && gc >= 0) g = make_int2(gc / G, gc % G);
                    sc = 0;         // v0:110
                    r_in = -0.0f;   // v0:109
                }
            }
            if (live) {
                int bx = clampi(b.x, 0, G - 1), by = clampi(b.y, 0, G - 1);
                sc += 1;            // v0:151, v3:225
                int ox, oy;
                decode_action(act, ox, oy);
                const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
                hits += transition_rule<VARIANT>(a, lays[tid * CELLS + tx * G + ty], ox, oy, tx, ty, sc, r_in, g.x, g.y, bx, by, r, dn) ? 1 : 0;
                b = make_int2(bx, by);
                was_done = dn ? 1 : 0;
                if (ro.reward_t) ro.reward_t[(size_t)t * a.n + e] = r;
                if (ro.done_t) ro.done_t[(size_t)t * a.n + e] = dn ? 1 : 0;
                ballflat[tid] = b.x * G + b.y;
                if (V3) goalflat[tid] = (g.x >= 0 && g.x < G && g.y >= 0 && g.y < G) ? g.x * G + g.y : -8;
            }
        }
        if (obs == nullptr) continue;                                                 // uniform
        __syncthreads();
        int le = (tid << 2) / CELLS, c = (tid << 2) - le * CELLS;
        const int dle = (LMAZE_BLOCK << 2) / CELLS, dc = (LMAZE_BLOCK << 2) - dle * CELLS;
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            int vals[4], l2 = le, c2 = c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = cell_bits<VARIANT>(lays[l2 * CELLS + c2]);
                v |= (ballflat[l2] == c2) ? LMAZE_OBS_BALL : 0;
                if (V3) v |= (goalflat[l2] == c2) ? LMAZE_OBS_GOAL : 0;
                vals[j] = v;
                if (++c2 == CELLS) { c2 = 0; ++l2; }
            }
            reinterpret_cast<int4*>(obs)[q] = make_int4(vals[0], vals[1], vals[2], vals[3]);
            le += dle; c += dc;
            if (c >= CELLS) { c -= CELLS; ++le; }
        }
        const int f = (nq << 2) + tid;                                                // ragged tail
        if (f < R) {
            const int l2 = f / CELLS, c2 = f - l2 * CELLS;
            int v = cell_bits<VARIANT>(lays[l2 * CELLS + c2]);
            v |= (ballflat[l2] == c2) ? LMAZE_OBS_BALL : 0;
            if (V3) v |= (goalflat[l2] == c2) ? LMAZE_OBS_GOAL : 0;
            obs[f] = v;
        }
        __syncthreads();
    }
    if (live && ro.T > 0) {
        a.ball[e] = b;
        if (V3 && autoreset) a.goal_rw[e] = g;
        a.step_count[e] = sc;
        a.reward[e] = r;
        a.done[e] = dn ? 1 : 0;
        if (hits && a.goal_count) a.goal_count[e] += hits;
    }
}

// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------

static size_t shared_lds_bytes(int G, bool specialised, int epb, bool with_marks) {
    const int cells = G * G;
    const int pat = (specialised && (G & 1)) ? 4 * cells : cells;
    // pattern + ballflat/goalflat/maskflag + layout bytes + spawn list (uint16 per cell) + ball / goal bit strings
    // (v3: two buffers x two planes x one string per group of envs)
    const int grp = (specialised && (G & 1)) ? 4 : 1;
    const size_t marks = (specialised && with_marks) ? (size_t)2 * 2 * (epb / grp) * ((grp * cells + 31) / 32) * 4 : 0;
    return (size_t)pat * 4 + 3 * (size_t)epb * 4 + (size_t)((cells + 15) & ~15) + (size_t)((cells * 2 + 15) & ~15) + marks;
}

int perenv_envs_per_block(int G, int layout_bytes) {
    const int cells = G * G;
    int epb = (layout_bytes / cells) & ~15;
    if (epb < 16) epb = 16;
    if (epb > LMAZE_BLOCK) epb = LMAZE_BLOCK;
    return epb;
}

static size_t perenv_lds_bytes(int G, int epb) {
    return (size_t)((epb * G * G + 15) & ~15) + (3 * (size_t)(epb + 1) + 2 * (size_t)epb) * 4;
}


// LDS bytes that make exactly `k` workgroups fit a CU's 160 KiB (midway between the k and
// k+1 thresholds, clear of the allocation granule).
static size_t lds_for_workgroups_per_cu(int k) {
    const size_t cap = 160 * 1024;
    return ((cap / k + cap / (k + 1)) / 2) & ~(size_t)255;
}

// Launch policy of the streaming (non-temporal) regime, measured on MI355X with
// tools/kbench.hip (one process, interleaved rounds), workgroups per CU = 8 (no cap) / 4 / 3 / 2:
//   1M x 11x11 (546 MB)   99-101 / 91-97 / 78-79 / 102 us     4M x 11x11   354 / 350 / 320 / 424 us
//   2M x 8x8   (614 MB)   99 / 97 / 92 / 88 us                128K x 32x32   99 / 96 / 92 / 106 us
//   1M x 12x12, 512K x 18x18: flat within 3 %
// i.e. up to 6.9 TB/s instead of 5.5 when only 12 waves per CU stream their 32-KiB chunks (more
// resident waves queue more non-temporal stores than the memory side drains efficiently; with
// plain stores the cap hurts).  The optimum is narrow and shifts with shape and device, so the
// caller can override it: LmazeParams.launch_hint bits 0-3 = workgroups per CU (0 = this
// default), which LmazeVecEnv.autotune() picks by timing real steps.
// ---- the default launch policy of the streaming regime is DATA: lmaze_policy_table.h, generated by tools/gen_policy.py
// from sweeps recorded on MI355X boxes (profiles/rNN/shape_sweep_*.jsonl), looked up by grid side, rules, fused reset and
// the nearest batch size in log2.  sel = the envs-per-workgroup selector of launch_hint bits 10-11 (0: the table's default
// one).  tests/test_gpu_policy_guard.py checks the table's choice against a small sweep on the box it runs on.
static bool specialised_grid(int g) { return g == 8 || g == 11 || g == 12 || g == 14 || g == 18 || g == 32; }

static const StepPolicyRow* step_policy_lookup(int g, int variant, bool auto_reset, int64_t n, int sel) {
    const StepPolicyRow* best = nullptr;
    long best_cost = 0;
    const int l2 = (int)lround(log2((double)(n < 1 ? 1 : n)) * 16.0);
    for (int i = 0; i < kStepPolicyRows; ++i) {
        const StepPolicyRow& r = kStepPolicy[i];
        if ((r.auto_reset != 0) != auto_reset) continue;
        if (sel == 0 ? !r.is_default : r.sel != sel) continue;
        if (specialised_grid(g) != specialised_grid(r.g)) continue;
        if (specialised_grid(g) && r.g != g) continue;
        // lexicographic: grid distance (unspecialised G only), then rules, then batch size
        const long cost = (long)abs(r.g - g) * 1000000 + (r.variant != variant ? 100000 : 0) + abs(r.log2n_x16 - l2);
        if (!best || cost < best_cost) { best = &r; best_cost = cost; }
    }
    return best;
}

// the selector (launch_hint bits 10-11) that names EPB envs per workgroup for this grid size; see launch_one
template <int GT, int EPB>
constexpr int sel_of() {
    if (GT == 11 || GT == 12) return EPB == 64 ? 1 : (EPB == 32 ? 2 : 3);
    if (GT == 14 || GT == 18) return EPB == 32 ? 1 : 2;
    if (GT == 8) return EPB == 128 ? 1 : 2;
    if (GT == 32) return EPB == 8 ? 1 : 2;
    return EPB == 256 ? 1 : (EPB == 64 ? 2 : 3);
}

template <int GT, int VARIANT, bool DO_STEP, int EPB>
static hipError_t launch_shared(const StepArgs& a, hipStream_t s) {
    const int64_t blocks = (a.n + EPB - 1) / EPB;
    size_t lds = shared_lds_bytes(a.grid, GT != 0, EPB, VARIANT == LMAZE_VARIANT_V3);
    const bool nt = a.obs != nullptr && (size_t)a.n * a.grid * a.grid * 4 > kNonTemporalObsBytes;
    // default (workgroups per CU, chunks per workgroup) of the streaming regime: the generated table (round 2 kept them
    // in a hand-edited if-chain here, retuned four times in its last hour; LAB_NOTES.md has that history)
    int def_cu = 3, def_m = 2;
    if (const StepPolicyRow* r = step_policy_lookup(a.grid, VARIANT, a.auto_reset != 0, a.n, sel_of<GT, EPB>())) {
        def_cu = r->per_cu;
        def_m = r->chunks;
    }
    int per_cu = a.launch_hint & 15;
    if (per_cu == 0 && nt) per_cu = def_cu;
    if (per_cu >= 1 && per_cu < 8) {
        const size_t want = lds_for_workgroups_per_cu(per_cu);
        if (want > lds) lds = want;
    }
    // chunks per workgroup (hint bits 4-7, 0 = default): the inputs of chunk k+1 are loaded while chunk k is
    // stored, and the per-workgroup LDS set-up is shared.
    int m = (a.launch_hint >> 4) & 15;
    if (m == 0) m = nt ? def_m : 1;
    const int64_t grid = (blocks + m - 1) / m;
    if (!grid_ok(grid)) return hipErrorInvalidConfiguration;
    if (a.info) {
        char name[96];
        snprintf(name, sizeof(name), "step_shared_kernel<%d, v%d, %s, %d, %s>", GT, VARIANT, DO_STEP ? "step" : "observe", EPB, nt ? "nt" : "plain");
        describe_launch(a.info, name, EPB, per_cu >= 1 && per_cu < 8 ? per_cu : 0, m, nt, grid, LMAZE_BLOCK, lds);
        return hipSuccess;
    }
    if (nt)
        hipLaunchKernelGGL((step_shared_kernel<GT, VARIANT, DO_STEP, EPB, true>), dim3((unsigned)grid),
                           dim3(LMAZE_BLOCK), lds, s, a);
    else
        hipLaunchKernelGGL((step_shared_kernel<GT, VARIANT, DO_STEP, EPB, false>), dim3((unsigned)grid),
                           dim3(LMAZE_BLOCK), lds, s, a);
    return hipGetLastError();
}

template <int G, int VARIANT, bool DO_STEP>
static hipError_t launch_perenv_wave(const StepArgs& a, hipStream_t s) {
    // One env per wave measured best at 1M x 32x32 (bench.py --workload c5): 0.873 ms per step
    // (0.775 of HBM peak, non-temporal stores) against 0.985 ms with four envs per wave and 0.96 ms
    // for the LDS-tiled kernel -- short-lived waves, as with the one-store-per-thread fill.
    StepArgs b = a;
    b.envs_per_block = 1;
    const int64_t per_block = (int64_t)b.envs_per_block * (LMAZE_BLOCK / 64);
    const int64_t blocks = (a.n + per_block - 1) / per_block;
    const bool nt = a.obs != nullptr && (size_t)a.n * G * G * 4 > kNonTemporalObsBytes;
    if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
    if (a.info) {
        char name[96];
        snprintf(name, sizeof(name), "step_perenv_wave_kernel<%d, v%d, %s, %s>", G, VARIANT, DO_STEP ? "step" : "observe", nt ? "nt" : "plain");
        describe_launch(a.info, name, (int)per_block, 0, 1, nt, blocks, LMAZE_BLOCK, 0);
        return hipSuccess;
    }
    if (nt)
        hipLaunchKernelGGL((step_perenv_wave_kernel<G, VARIANT, DO_STEP, true>), dim3((unsigned)blocks),
                           dim3(LMAZE_BLOCK), 0, s, b);
    else
        hipLaunchKernelGGL((step_perenv_wave_kernel<G, VARIANT, DO_STEP, false>), dim3((unsigned)blocks),
                           dim3(LMAZE_BLOCK), 0, s, b);
    return hipGetLastError();
}

template <int GT, int VARIANT, bool DO_STEP>
static hipError_t launch_perenv(const StepArgs& a, hipStream_t s) {
    // G*G a multiple of 256: the register-tiled one-wave-per-env kernel
    if constexpr (GT == 32) return launch_perenv_wave<32, VARIANT, DO_STEP>(a, s);
    if constexpr (GT == 0) {
        if (a.grid == 16) return launch_perenv_wave<16, VARIANT, DO_STEP>(a, s);
        if (a.grid == 48) return launch_perenv_wave<48, VARIANT, DO_STEP>(a, s);
        if (a.grid == 64) return launch_perenv_wave<64, VARIANT, DO_STEP>(a, s);
    }
    StepArgs b = a;
    // about 8 KiB of layouts = 32 KiB of planes per workgroup (round 1: 16 KiB; 1M x 11x11 / 12x12, 512K x 18x18 with
    // non-temporal stores: 107 / 125 / 134 us with 16 KiB, 102-103 / 120-122 / 133-140 with 8)
    b.envs_per_block = perenv_envs_per_block(a.grid, 8192);
    const int64_t blocks = (a.n + b.envs_per_block - 1) / b.envs_per_block;
    if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
    if (a.info) {
        char name[96];
        snprintf(name, sizeof(name), "step_perenv_kernel<%d, v%d, %s>", GT, VARIANT, DO_STEP ? "step" : "observe");
        describe_launch(a.info, name, b.envs_per_block, 0, 1, false, blocks, LMAZE_BLOCK, perenv_lds_bytes(a.grid, b.envs_per_block));
        return hipSuccess;
    }
    hipLaunchKernelGGL((step_perenv_kernel<GT, VARIANT, DO_STEP>), dim3((unsigned)blocks), dim3(LMAZE_BLOCK),
                       perenv_lds_bytes(a.grid, b.envs_per_block), s, b);
    return hipGetLastError();
}

// the table's default envs-per-workgroup selector for this launch (streaming regime), or `fallback`
template <int VARIANT>
static int table_sel(const StepArgs& a, int fallback) {
    const StepPolicyRow* r = step_policy_lookup(a.grid, VARIANT, a.auto_reset != 0, a.n, 0);
    return r ? r->sel : fallback;
}

// Envs per workgroup for the shared-layout kernel: about 32 KiB of observation per workgroup
// (8 sixteen-byte stores per lane).  Measured at G=11 on MI355X (tools/kbench.hip): 64 envs
// (31 KiB) 90-91 us per 1M-env step, 128/256 envs 96-101 us, 32 envs 120 us.
template <int GT, int VARIANT, bool DO_STEP>
static hipError_t launch_one(const StepArgs& a, int layout_mode, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (layout_mode != LMAZE_LAYOUT_SHARED) return launch_perenv<GT, VARIANT, DO_STEP>(a, s);
    if constexpr (GT == 8) {
        // small batches (the planes stay on-die): the wave-autonomous kernel, no LDS, no barrier
        const bool small = a.obs == nullptr || (size_t)a.n * 64 * 4 <= kNonTemporalObsBytes;
        if (small && a.mask == nullptr && (a.launch_hint & 0x100) == 0) {
            // envs per wave: launch_hint bits 4-7 = 1: 64, 2: 32, 3: 16 (0 = default)
            // Defaults, measured at 65 536 envs under hipGraph (us per step, two boxes): 64 envs per wave x 4 waves per
            // workgroup -- one 256-thread workgroup per CU -- 5.35-5.53; 64 x 1 5.6-5.9; 32 x 1 / 2 / 4 5.72 / 5.72-5.87 /
            // 5.89; 64 x 2 5.9-6.1.  Smaller batches keep single-wave workgroups so that they spread over the CUs.
            int code = (a.launch_hint >> 4) & 15;
            if (code < 1 || code > 3) code = a.n >= 65536 ? 1 : 2;
            const int epw = 128 >> code;
            int wpb = a.launch_hint & 15;                   // bits 0-3: waves per workgroup, 1 / 2 / 4 (0 = default)
            if (wpb != 1 && wpb != 2 && wpb != 4) wpb = a.n >= 65536 ? 4 : 1;
            const int64_t waves = (a.n + epw - 1) / epw;
            const int64_t blocks = (waves + wpb - 1) / wpb;
            if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
            const dim3 grid((unsigned)blocks), block(64 * wpb);
            if (a.info) {
                char name[96];
                snprintf(name, sizeof(name), "step_shared_wave8_kernel<v%d, %s, %d>", VARIANT, DO_STEP ? "step" : "observe", epw);
                describe_launch(a.info, name, epw * wpb, 0, 1, false, blocks, 64 * wpb, 0);
                return hipSuccess;
            }
            if (code == 1) hipLaunchKernelGGL((step_shared_wave8_kernel<VARIANT, DO_STEP, 64>), grid, block, 0, s, a);
            else if (code == 2) hipLaunchKernelGGL((step_shared_wave8_kernel<VARIANT, DO_STEP, 32>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((step_shared_wave8_kernel<VARIANT, DO_STEP, 16>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        // large 8x8 batches: launch_hint bits 10-11: 1: 128 envs per workgroup, 2: 64 (16 KiB of planes).  2M envs: 64 envs at
        // 4 / 5 / 6 / 8 per CU 89.0 / 91.7 / 91.8 / 91.7 us, 128 envs at (2, 1) 92.6, at 3-8 per CU 102-105.
        int sel = (a.launch_hint >> 10) & 3;
        if (sel == 0) sel = table_sel<VARIANT>(a, a.obs != nullptr ? 2 : 1);
        if (sel == 2) return launch_shared<GT, VARIANT, DO_STEP, 64>(a, s);
        return launch_shared<GT, VARIANT, DO_STEP, 128>(a, s);
    } else if constexpr (GT == 11 || GT == 12) {
        // launch_hint bits 10-11: envs per workgroup, 1: 64, 2: 32 (0 = default).  Round 2, once the set-up was one global
        // round trip: v0 11x11 1M, 32 envs (15 KiB of planes) per workgroup, uncapped, one chunk 76.1-76.6 us on every
        // placement of the observation buffer tried, against 78.6-80.9 for 64 envs at (3, 2) / (3, 1) (16 envs: 90+);
        // with the fused reset (after its set-up lost the compacted spawn list and the early wait) 32 envs uncapped 78.8-83.5
        // on two boxes, 64 envs at (3, 2) 81 / 95-97, at (3, 1) 81-84.
        int sel = (a.launch_hint >> 10) & 3;
        const bool streaming = a.obs != nullptr && (size_t)a.n * GT * GT * 4 > kNonTemporalObsBytes;
        // 12x12 (1M envs, 643 MB): 16 envs (9 KiB) per workgroup, uncapped 97.1-97.4 us, (5, 2) 96.0, against 104-105 for 64
        // envs at (2, 1) and 111-119 for nearly everything else; with the fused reset 103-104 against 109.  (3: 16 envs.)
        if (sel == 0) {
            if (GT == 11) sel = streaming ? 2 : 1;      // the render-only launches (lmaze_observe) write the same stream
            else sel = streaming ? 3 : 1;
            if (streaming) sel = table_sel<VARIANT>(a, sel);
        }
        if (sel == 2) return launch_shared<GT, VARIANT, DO_STEP, 32>(a, s);
        if (sel == 3) return launch_shared<GT, VARIANT, DO_STEP, 16>(a, s);
        return launch_shared<GT, VARIANT, DO_STEP, 64>(a, s);
    } else if constexpr (GT == 14 || GT == 18) {
        // launch_hint bits 10-11: envs per workgroup, 1: 32, 2: 16 (0 = default).  1M x 14x14 v0 (861 MB): 16 envs (12 KiB
        // of planes) at 5 workgroups per CU 120.5 us, at 4 / 6 / uncapped 134 / 142 / 142, against 150-161 for every
        // policy of 32 envs but (2, 2) (132.5); 512K x 18x18 v3: 16 envs at (4, 1) or (3, 2) 104.3-104.5 against 109.5
        // for 32 at (2, 1) -- but 118-134 one step to either side, so there it stays a tuner candidate.
        int sel = (a.launch_hint >> 10) & 3;
        const bool streaming = a.obs != nullptr && (size_t)a.n * GT * GT * 4 > kNonTemporalObsBytes;
        if (sel == 0) sel = ((GT == 14 || VARIANT == LMAZE_VARIANT_V3) && streaming) ? 2 : 1;
        if (((a.launch_hint >> 10) & 3) == 0 && streaming) sel = table_sel<VARIANT>(a, sel);   // v3 18x18 since its render went to bit strings: 16 envs at 4-5 per CU 109 us (with the fused reset 5-8 per CU 103-107) against 112-120
        if (sel == 2) return launch_shared<GT, VARIANT, DO_STEP, 16>(a, s);
        return launch_shared<GT, VARIANT, DO_STEP, 32>(a, s);
    } else if constexpr (GT == 32) {
        // launch_hint bits 10-11: 1: 8 envs per workgroup (32 KiB of planes), 2: 4 envs (16 KiB).  128K / 512K envs, us per
        // step: 4 envs at (5, 1) 84.0 / 337 and at (4, 1) 85.0 / 323 against 96 / 363 for 8 envs at (2, 1); with the fused
        // reset 4 envs at (5, 1) 88.6 / 342, at (6, 1) 81.8 / 313, against 102 / 403.
        int sel = (a.launch_hint >> 10) & 3;
        const bool streaming = a.obs != nullptr && (size_t)a.n * GT * GT * 4 > kNonTemporalObsBytes;
        if (sel == 0) sel = streaming ? table_sel<VARIANT>(a, 2) : 1;
        if (sel == 2) return launch_shared<GT, VARIANT, DO_STEP, 4>(a, s);
        return launch_shared<GT, VARIANT, DO_STEP, 8>(a, s);
    } else {  // unspecialised G: three sizes cover [3, 64]; launch_hint bits 10-11: 1: 256, 2: 64, 3: 16 envs per workgroup
        // Round 2 (1M envs at G = 9, 10, 13; 512K at 16, 20; 256K at 27; us per step, old -> new default): 64.9 -> 56.4,
        // 88.0 -> 76.2, 120.7 (unchanged), 118.9 -> 90.0 (16 envs at (5, 2): 81.7), 169.7 -> 148.1, 155.3 -> 152.
        int sel = (a.launch_hint >> 10) & 3;
        if (sel == 0) {
            sel = a.grid >= 15 ? 3 : (a.grid >= 5 ? 2 : 1);
            if (a.obs != nullptr && (size_t)a.n * a.grid * a.grid * 4 > kNonTemporalObsBytes && a.grid >= 5) sel = table_sel<VARIANT>(a, sel);
        }
        if (sel == 3) return launch_shared<GT, VARIANT, DO_STEP, 16>(a, s);
        if (sel == 2) return launch_shared<GT, VARIANT, DO_STEP, 64>(a, s);
        return launch_shared<GT, VARIANT, DO_STEP, 256>(a, s);
    }
}

// Grid sizes the kernels are specialised for: the reference's shipped sizes (12, 14, 18)
// and BASELINE.json's (8, 11, 32); any other G in [3, 64] takes the GT = 0 instantiation.
template <int VARIANT, bool DO_STEP>
static hipError_t dispatch_grid(const StepArgs& a, int layout_mode, hipStream_t s) {
    switch (a.grid) {
        case 8:  return launch_one<8, VARIANT, DO_STEP>(a, layout_mode, s);
        case 11: return launch_one<11, VARIANT, DO_STEP>(a, layout_mode, s);
        case 12: return launch_one<12, VARIANT, DO_STEP>(a, layout_mode, s);
        case 14: return launch_one<14, VARIANT, DO_STEP>(a, layout_mode, s);
        case 18: return launch_one<18, VARIANT, DO_STEP>(a, layout_mode, s);
        case 32: return launch_one<32, VARIANT, DO_STEP>(a, layout_mode, s);
        default: return launch_one<0, VARIANT, DO_STEP>(a, layout_mode, s);
    }
}

hipError_t launch_step(int variant, bool do_step, const StepArgs& a, int layout_mode, hipStream_t s);

#ifndef LMAZE_U8_DEFAULT_SEL
#define LMAZE_U8_DEFAULT_SEL 1
#endif
#ifndef LMAZE_U8_DEFAULT_CHUNKS
#define LMAZE_U8_DEFAULT_CHUNKS 2
#endif
// narrow (uint8) observation, shared layout.  launch_hint bits 10-11: envs per workgroup, 1: 256, 2: 128, 3: 64 (0 = default)
template <int EPB>
static hipError_t launch_step_u8_epb(int variant, bool do_step, const StepArgs& a, hipStream_t s) {
    const int cells = a.grid * a.grid, pw = (2 * cells + 16 + 3) >> 2;
    const size_t lds = ((size_t)4 * pw * 4 + 3 * (size_t)(EPB + 1) * 4 + (size_t)((cells + 1) & ~1) * 2 + (size_t)cells + 15) & ~(size_t)15;
    // launch_hint bits 4-7: chunks of EPB envs per workgroup (0 = default)
    int m = (a.launch_hint >> 4) & 15;
    if (m == 0) m = LMAZE_U8_DEFAULT_CHUNKS;
    const int64_t nchunks = (a.n + EPB - 1) / EPB;
    const int64_t blocks = (nchunks + m - 1) / m;
    if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
    if (a.info) {
        char name[96];
        snprintf(name, sizeof(name), "step_shared_u8_kernel<v%d, %s, %d>", variant, do_step ? "step" : "observe", EPB);
        describe_launch(a.info, name, EPB, 0, m, false, blocks, LMAZE_BLOCK, lds);
        return hipSuccess;
    }
    const dim3 grid((unsigned)blocks), block(LMAZE_BLOCK);
    StepArgs b = a;
    // more than 64 MiB of planes: streamed (1M x 11x11, 127 MB: 31.2 us per step against 33.3 with plain stores)
    b.launch_hint = (a.launch_hint & ~0x100) | ((size_t)a.n * cells > ((size_t)64 << 20) ? 0x100 : 0);
    if (variant == LMAZE_VARIANT_V3) {
        if (do_step) hipLaunchKernelGGL((step_shared_u8_kernel<LMAZE_VARIANT_V3, true, EPB>), grid, block, lds, s, b);
        else hipLaunchKernelGGL((step_shared_u8_kernel<LMAZE_VARIANT_V3, false, EPB>), grid, block, lds, s, b);
    } else {
        if (do_step) hipLaunchKernelGGL((step_shared_u8_kernel<LMAZE_VARIANT_V0, true, EPB>), grid, block, lds, s, b);
        else hipLaunchKernelGGL((step_shared_u8_kernel<LMAZE_VARIANT_V0, false, EPB>), grid, block, lds, s, b);
    }
    return hipGetLastError();
}

hipError_t launch_step_u8(int variant, bool do_step, const StepArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    int sel = (a.launch_hint >> 10) & 3;
    if (sel == 0) sel = LMAZE_U8_DEFAULT_SEL;
    if (sel == 1) return launch_step_u8_epb<256>(variant, do_step, a, s);
    if (sel == 2) return launch_step_u8_epb<128>(variant, do_step, a, s);
    return launch_step_u8_epb<64>(variant, do_step, a, s);
}

// T steps: ONE launch (shared layouts: rollout_shared_wave8_kernel for on-die 8x8, rollout_shared_kernel otherwise; per-env
// layouts: rollout_perenv_kernel); T launches of the step kernel only for T = 1 or when launch_hint bit 8 forces streaming stores, step t with the action row t, epoch + t and, when given, the per-step reward / done rows copied out.
hipError_t launch_rollout(int variant, const StepArgs& a0, int layout_mode, const int32_t* actions, int32_t T, float* reward_t,
                          uint8_t* done_t, hipStream_t s) {
    if (T <= 0 || a0.n == 0) return hipSuccess;
    const bool on_die8 = layout_mode == LMAZE_LAYOUT_SHARED && a0.grid == 8 &&
                         (a0.obs == nullptr || (size_t)a0.n * 64 * 4 <= kNonTemporalObsBytes) && (a0.launch_hint & 0x100) == 0;
    if (on_die8) {
        RolloutArgs ro{actions, reward_t, done_t, T};
        const int epw = 64, wpb = a0.n >= 65536 ? 4 : 1;
        const int64_t waves = (a0.n + epw - 1) / epw, blocks = (waves + wpb - 1) / wpb;
        if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
        if (a0.info) {
            char name[96];
            snprintf(name, sizeof(name), "rollout_shared_wave8_kernel<v%d, %d> T=%d", variant, epw, T);
            describe_launch(a0.info, name, epw * wpb, 0, 1, false, blocks, 64 * wpb, 0);
            return hipSuccess;
        }
        if (variant == LMAZE_VARIANT_V3)
            hipLaunchKernelGGL((rollout_shared_wave8_kernel<LMAZE_VARIANT_V3, 64>), dim3((unsigned)blocks), dim3(64 * wpb), 0, s, a0, ro);
        else
            hipLaunchKernelGGL((rollout_shared_wave8_kernel<LMAZE_VARIANT_V0, 64>), dim3((unsigned)blocks), dim3(64 * wpb), 0, s, a0, ro);
        return hipGetLastError();
    }
    // any other shared-layout batch: one launch of rollout_shared_kernel, a workgroup per 16 / 32 / 64 envs so that small
    // batches still fill the chip.  Streaming sizes too: the planes of every step still go out to HBM, but the state stays
    // in registers and there is no per-step set-up -- 1M x 11x11 75.4 us per step (fused reset 75.6) against 86.4 (92.5) as T
    // launches from inside this call, 4M envs 297.8 against 329.7 (profiles/r03/rollout_streaming.txt)
    const bool on_die = layout_mode == LMAZE_LAYOUT_SHARED && T > 1 && (a0.launch_hint & 0x100) == 0;
    if (on_die) {
        RolloutArgs ro{actions, reward_t, done_t, T};
        StepArgs a = a0;
        a.envs_per_block = a0.n >= 65536 ? 64 : (a0.n >= 16384 ? 32 : 16);
        // Batches whose planes do not fit the caches: few enough envs per workgroup that what the resident workgroups of an XCD
        // rewrite step after step (8 per CU x 32 CUs x envs x 4 G^2 bytes) stays inside its 4-MiB L2 -- then most of the
        // intermediate planes never travel to DRAM.  1M x 11x11: 64 envs 77.7 us per step, 32: 78.6, 16: 51.0, 8: 64.2, 4: 109;
        // 512K x 18x18: 64: 89.2, 16: 98.4, 8: 74.2, 4: 76.2; 256K x 32x32 (no size fits): 146 / 140 / 133.7 / 141.6
        // (profiles/r03/rollout_envs_per_workgroup.txt).  launch_hint bits 12-14 = k > 0 ask for 4 << (k - 1) envs.
        // (Only beyond the Infinity Cache: 262 144 x 11x11, 127 MB of planes, runs 9.9 us per step at 64 envs and 12.9 at 16.)
        if ((size_t)a0.n * a0.grid * a0.grid * 4 > kNonTemporalObsBytes) {
            int fit = 64;
            while (fit > 8 && (size_t)fit * a0.grid * a0.grid * 4 > 12288) fit >>= 1;
            if (fit < a.envs_per_block) a.envs_per_block = fit;
        }
        if ((a0.launch_hint >> 12) & 7) a.envs_per_block = 4 << (((a0.launch_hint >> 12) & 7) - 1);
        const int cells = a0.grid * a0.grid;
        const size_t lds = (size_t)cells * 4 + 2 * (size_t)a.envs_per_block * 4 + (size_t)((cells + 15) & ~15) + (size_t)((cells * 2 + 15) & ~15);
        const int64_t blocks = (a0.n + a.envs_per_block - 1) / a.envs_per_block;
        if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
        if (a0.info) {
            char name[96];
            snprintf(name, sizeof(name), "rollout_shared_kernel<v%d> T=%d", variant, T);
            describe_launch(a0.info, name, a.envs_per_block, 0, 1, false, blocks, LMAZE_BLOCK, lds);
            return hipSuccess;
        }
        if (variant == LMAZE_VARIANT_V3)
            hipLaunchKernelGGL((rollout_shared_kernel<LMAZE_VARIANT_V3>), dim3((unsigned)blocks), dim3(LMAZE_BLOCK), lds, s, a, ro);
        else
            hipLaunchKernelGGL((rollout_shared_kernel<LMAZE_VARIANT_V0>), dim3((unsigned)blocks), dim3(LMAZE_BLOCK), lds, s, a, ro);
        return hipGetLastError();
    }
    // per-env layouts, any size: the layouts are read ONCE per rollout instead of once per step (1M x 32x32: 838 -> 722 us per
    // step, 1M x 11x11 103 -> 100.5, with the fused reset 114 -> 105.7, 512K x 18x18 142 -> 114; only 8 192 x 32x32, where the
    // register-tiled step kernel is neither launch- nor layout-bound, is 3 % slower: 10.2 -> 10.5;
    // profiles/r03/rollout_streaming_perenv.txt, rollout_sizes_perenv.txt)
    const bool on_die_perenv = layout_mode == LMAZE_LAYOUT_PER_ENV && T > 1 && (a0.launch_hint & 0x100) == 0;
    if (on_die_perenv) {
        RolloutArgs ro{actions, reward_t, done_t, T};
        StepArgs a = a0;
        const int cells = a0.grid * a0.grid;
        int epb = a0.n >= 65536 ? 64 : (a0.n >= 16384 ? 32 : 16);
        while (epb > 4 && (size_t)epb * cells > (size_t)32 << 10) epb >>= 1;         // at most 32 KiB of layouts per workgroup
        if ((size_t)a0.n * cells * 4 > kNonTemporalObsBytes)                         // beyond the caches: planes that fit the L2s, as above
            while (epb > 8 && (size_t)epb * cells * 4 > 12288) epb >>= 1;
        if ((a0.launch_hint >> 12) & 7) epb = 4 << (((a0.launch_hint >> 12) & 7) - 1);
        if (epb > 64) epb = 64;                                                      // every env's lane sits in wave 0
        a.envs_per_block = epb;
        const size_t lds = 2 * (size_t)epb * 4 + (((size_t)epb * cells + 15) & ~(size_t)15);
        const int64_t blocks = (a0.n + epb - 1) / epb;
        if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
        if (a0.info) {
            char name[96];
            snprintf(name, sizeof(name), "rollout_perenv_kernel<v%d> T=%d", variant, T);
            describe_launch(a0.info, name, epb, 0, 1, false, blocks, LMAZE_BLOCK, lds);
            return hipSuccess;
        }
        if (variant == LMAZE_VARIANT_V3)
            hipLaunchKernelGGL((rollout_perenv_kernel<LMAZE_VARIANT_V3>), dim3((unsigned)blocks), dim3(LMAZE_BLOCK), lds, s, a, ro);
        else
            hipLaunchKernelGGL((rollout_perenv_kernel<LMAZE_VARIANT_V0>), dim3((unsigned)blocks), dim3(LMAZE_BLOCK), lds, s, a, ro);
        return hipGetLastError();
    }
    for (int32_t t = 0; t < T; ++t) {
        StepArgs a = a0;
        a.action = actions + (size_t)t * a0.n;
        a.epoch = a0.epoch + (uint64_t)t;
        hipError_t rc = launch_step(variant, true, a, layout_mode, s);
        if (rc != hipSuccess || a0.info) return rc;
        if (reward_t && (rc = hipMemcpyAsync(reward_t + (size_t)t * a0.n, a0.reward, (size_t)a0.n * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess) return rc;
        if (done_t && (rc = hipMemcpyAsync(done_t + (size_t)t * a0.n, a0.done, (size_t)a0.n, hipMemcpyDeviceToDevice, s)) != hipSuccess) return rc;
    }
    return hipSuccess;
}

hipError_t launch_step(int variant, bool do_step, const StepArgs& a, int layout_mode, hipStream_t s) {
    if (variant == LMAZE_VARIANT_V3)
        return do_step ? dispatch_grid<LMAZE_VARIANT_V3, true>(a, layout_mode, s)
                       : dispatch_grid<LMAZE_VARIANT_V3, false>(a, layout_mode, s);
    return do_step ? dispatch_grid<LMAZE_VARIANT_V0, true>(a, layout_mode, s)
                   : dispatch_grid<LMAZE_VARIANT_V0, false>(a, layout_mode, s);
}

}  // namespace lmaze
