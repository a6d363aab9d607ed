// lmaze_foveal.hip -- the foveal variants of the step path (5x5 window observations), gfx950.
//
//   v1 = gym_lmaze/envs/lmaze_env_v1.py:114-200   4-neighbour move, two reward streams, foveal goal
//   v2 = gym_lmaze/envs/lmaze_env_v2.py:127-225   25-way teleport inside the fovea, 5 layouts
//   v4 = gym_lmaze/envs/lmaze_env_v4.py:167-272   v2 + float visit-map plane (whole-plane halving)
//
//   v5/v6 = lmaze_env_v5.py:158-292 (plannerStep + step), lmaze_env_v6.py:505-523 (safeFovealGoal)
//
// Same two-phase shape as lmaze_step.hip: one lane per env runs the transition against the
// layout table held in LDS and leaves the env's observation in LDS as 25-bit plane masks; then the
// workgroup's lanes stripe its contiguous observation range float[envs*C*25] with 16-byte stores.
// v4-v6 add a middle phase on the visit map, kept CLOCK-RELATIVE and TILED (include/lmaze.h "The visit map"): the
// reference halves the whole G x G plane on every update (v4:211-214, v5:313-318); here that is `clock += 1` and a
// step only gathers the 2 x 2 tiles (4 x 4 cells, 64 B each) under the window it shows, adds, and writes them back.
// HBM bytes per env-step: v1 454, v2 545, v4 about 1 000 instead of round 2's 3 337.
#include <cstdio>
#include <cstring>

#include "lmaze_common.h"
#include "lmaze_visit.h"

// Timing decomposition (tools/foveal_decompose.py): a build with -DLMAZE_EXPERIMENT -- never the shipped
// one -- reads bits 16-23 of launch_hint as switches that turn phases off (results are garbage then; only the time
// counts): 1 no set-up, 2 plain instead of non-temporal observation stores, 4 no observation stores, 8 no phase 1,
// 16 stores as interleaved 4-KiB pieces, 32 no visit-map phase, 64 / 128 non-temporal visit-map stores / loads.
#ifdef LMAZE_EXPERIMENT
#define LMAZE_WARM_V4(args) ((((args).p.launch_hint >> 16) & 1024) != 0)
#define LMAZE_XP(args, mask) ((((args).p.launch_hint >> 16) & (mask)) != 0)   // round 3: 256 no per-cell work on gathered tiles, 512 no "previous" window tiles
#else
#define LMAZE_WARM_V4(args) false
#define LMAZE_XP(args, mask) false
#endif

namespace lmaze {

constexpr int FOV = LMAZE_FOVEA;
constexpr int W25 = FOV * FOV;

enum FovealMode { FM_STEP = 0, FM_RESET = 1, FM_SETGOAL = 2, FM_PLANNER = 3 };

struct FovealArgs {
    LmazeFovealParams p;
    LmazeFovealBuffers b;
    const uint8_t* layouts;
    const int32_t* action;  // step: action ids; setgoal: ij[N,2]
    const int32_t* goal2;   // v5 two-level step: planner goals (plannerStep of the envs that enter with localDone / done)
    const uint8_t* mask;
    int64_t n;
    int32_t place;
    uint64_t seed, epoch;
    int64_t env_base;
    const uint64_t* epoch_in;  // fused auto-reset from a captured graph: device-resident epoch (lmaze_common.h)
    uint64_t* epoch_out;
    int32_t nt;             // non-temporal observation stores (set by the launcher)
    int32_t auto_reset;     // step: an env whose done flag is set on entry is reset first (v1, v2, v4)
    LaunchInfo* info;       // host pointer; non-null: describe the launch instead of queueing it (lmaze_describe_foveal_step)
};

struct EnvRec {           // one env after its transition (registers only; phase 1 turns it into plane masks)
    int16_t cx, cy;       // centre of the current window (ball after the move)
    int16_t px, py;       // centre of the "previous" window
    int16_t gx, gy;       // goal (v2/v4) or foveal goal (v1 local view)
    int16_t lid;          // row of the layout table
    int16_t action;       // v2/v4 action plane (-1: none); v1: 1 = local view, 0 = global view
    int32_t skip;         // env untouched by this call: neither state nor obs are written
    int32_t flat;         // v1 local view: flat index of the one-hot goal (numpy wrap applied), -1 none
    int16_t b0x, b0y;     // v5/v6 local observation: ball, previous ball, fovea_1 (v5:364-365)
    int16_t b1x, b1y;
    int16_t f1x, f1y;
    int16_t upd;          // v5/v6: localDone -> this call halves the visit map (v5:313-318)
    int16_t pad;
};

// Placement on row masks.  rows[x] has bit y set when interior cell (x, y) is accepted; accepted cells are
// ranked in row-major order (the order of the reference's own scan over the grid).

// accepted cells of the goal and of the ball mask, one pass.  Unrolled by 4 and no more: the loop is a chain of LDS round
// trips in a lane that a whole wave waits for (some lane of most waves resets at steady state), so it wants several
// reads in flight, but with G known at compile time a FULL unroll keeps a layout's row masks live and the fused-reset
// instantiation then needs 115 VGPRs (4 waves per SIMD instead of 7) for the whole kernel
template <int U>
__device__ __forceinline__ void mask_counts(const uint64_t* goal_rows, const uint64_t* ball_rows, int G, int& cg, int& cb) {
    cg = 0; cb = 0;
#pragma unroll U
    for (int x = 1; x <= G - 2; ++x) { cg += __popcll(goal_rows[x]); cb += __popcll(ball_rows[x]); }
}

// k-th accepted cell (0-based) as x*G + y, or -1, of the mask with cell `hole` (x*G + y; negative: none) taken out.
// No early exit, so that the row reads pipeline (see mask_counts).  The hole is tested as "hole - x*G in [0, G)" so that
// nothing but the cell index itself stays live across the loop.
template <int U>
__device__ __forceinline__ int mask_kth(const uint64_t* rows, int G, int k, int hole) {
    int xr = -1, kk = 0, acc = 0;
#pragma unroll U
    for (int x = 1; x <= G - 2; ++x) {
        uint64_t m = rows[x];
        const unsigned hy = (unsigned)(hole - x * G);
        if (hy < (unsigned)G) m &= ~(1ull << hy);
        const int c = __popcll(m);
        if (xr < 0 && k < acc + c) { xr = x; kk = k - acc; }
        acc += c;
    }
    if (xr < 0) return -1;
    uint64_t m = rows[xr];
    const unsigned hy = (unsigned)(hole - xr * G);
    if (hy < (unsigned)G) m &= ~(1ull << hy);
    uint32_t h = (uint32_t)m;
    int base = 0;
    const int cl = __popc(h);
    if (kk >= cl) { kk -= cl; h = (uint32_t)(m >> 32); base = 32; }
    for (; kk > 0; --kk) h &= h - 1;
    return xr * G + base + (__ffs((int)h) - 1);
}

// reset() placement of v2/v4/v5/v6 on one layout (v2:277-296): goal uniform over interior cells that are
// not 'W' and not 'S'; ball uniform over interior cells that are not 'W', not 'X' and not the goal.  Accepted cells
// are ranked row-major (the order of the reference's own scan); "not the goal" = the goal's bit taken out of the
// ball mask, which leaves the ranking of every other cell as the reference's list has it.
template <int U>
__device__ __forceinline__ void place_goal_ball(const uint64_t* goal_rows, const uint64_t* ball_rows, int G, uint4 d,
                                                int& goal_cell, int& ball_cell) {
    goal_cell = -1;
    ball_cell = -1;
    int cg, cb;
    mask_counts<U>(goal_rows, ball_rows, G, cg, cb);
    if (cg > 0) goal_cell = mask_kth<U>(goal_rows, G, (int)__umulhi(d.x, (uint32_t)cg), -1);
    if (goal_cell >= 0 && ((ball_rows[goal_cell / G] >> (goal_cell % G)) & 1ull)) --cb;   // a 'B' goal cell leaves the ball's list
    if (cb > 0) ball_cell = mask_kth<U>(ball_rows, G, (int)__umulhi(d.y, (uint32_t)cb), goal_cell);
}

// numpy index semantics on an axis of 5: -5..-1 wrap, anything else outside 0..4 raises (-> -1)
__device__ __forceinline__ int wrap5(int i) {
    if (i >= 0 && i < FOV) return i;
    if (i < 0 && i >= -FOV) return i + FOV;
    return -1;
}

// 25-bit mask (bit 5*i+j) of a 5x5 window centred on (cx, cy) over a plane given as one 64-bit row
// mask per layout row (bit y = cell (x, y) is set); cells outside the array read 0
__device__ __forceinline__ uint32_t window_bits(const uint64_t* rows, int G, int cx, int cy) {
    uint32_t m = 0;
    const int y0 = cy - 2;
#pragma unroll
    for (int i = 0; i < FOV; ++i) {
        const int x = cx - 2 + i;
        const uint64_t b = (x >= 0 && x < G) ? rows[x] : 0ull;
        const uint32_t w = (uint32_t)(y0 >= 0 ? (b >> y0) : (b << -y0)) & 31u;
        m |= w << (FOV * i);
    }
    return m;
}

// bit of cell (tx, ty) inside the window centred on (cx, cy), 0 if it is outside the window
__device__ __forceinline__ uint32_t onehot_bits(int tx, int ty, int cx, int cy) {
    const int i = tx - cx + 2, j = ty - cy + 2;
    return (i >= 0 && i < FOV && j >= 0 && j < FOV) ? (1u << (FOV * i + j)) : 0u;
}

// OR the 25-bit plane m into a bit string at bit offset off (LDS atomics: neighbouring envs share words)
__device__ __forceinline__ void put_bits(uint32_t* bits, int off, uint32_t m) {
    const int w = off >> 5, sh = off & 31;
    atomicOr(&bits[w], m << sh);
    if (sh > 32 - W25) atomicOr(&bits[w + 1], m >> (32 - sh));
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4, gfx950): no register destination.
// `lds_wave_base` is WAVE-UNIFORM: lane l's bytes land at lds_wave_base + 16 l whatever the exec mask.
__device__ __forceinline__ void lds_dma16(const uint32_t* src, uint32_t* lds_wave_base) {
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    __builtin_amdgcn_global_load_lds((gvoid*)src, (lvoid*)lds_wave_base, 16, 0, 0);
}

// four consecutive floats (0.0f / 1.0f) from nibble q of a bit string
__device__ __forceinline__ void nibble_floats(const uint32_t* bits, int q, float (&v)[4]) {
    const uint32_t nib = bits[q >> 3] >> ((q & 7) << 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __uint_as_float((0u - ((nib >> k) & 1u)) & 0x3f800000u);
}

// The visit map in its clock-relative frame: the bit-pattern arithmetic lives in lmaze_visit.h (shared with the host-side
// property test of the CPU suite); here only the tile geometry.
constexpr int VISIT_BIAS = LMAZE_VISIT_BIAS, VISIT_RENORM = LMAZE_VISIT_RENORM;
constexpr int VT = 4;                // tile side; a tile is 16 floats = 64 bytes
// Behind an env batch's tiles the visit buffer holds one record of VPC words per env: the true values of the 5x5 window
// the observation shows as "previous" (retStatelast: v4:239,259, v5:322-346), word 25 = a tag naming the centre they
// belong to.  v5/v6 show that window unchanged for up to ten steps and v4 shows last step's current window, so it is
// kept as 112 contiguous bytes instead of being gathered from up to four more tiles every step.
constexpr int VPC = 28;
// Which centre the record belongs to rides in the upper bits of the env's visit_clock word (bits 0-7 the clock, bit 8
// "record valid", bits 9-15 / 16-22 the centre): one coalesced load in phase 1 tells whether the record serves this call.
__host__ __device__ __forceinline__ int visit_tag(int x, int y) { return 0x100 | ((x & 0x7f) << 9) | ((y & 0x7f) << 16); }

__host__ __device__ __forceinline__ int visit_tiles(int G) { return (G + VT - 1) / VT; }
__device__ __forceinline__ uint32_t visit_true(uint32_t bits, int E) { return lmaze_visit_true(bits, E); }
__device__ __forceinline__ uint32_t visit_add(uint32_t bits, int E) { return lmaze_visit_add(bits, E); }

// GT = grid side known at compile time (14 and 18, the reference's sizes; 0: read it from the params):
// the visit-map stream divides by G for every cell, which is only cheap with a constant
// AR = fused auto-reset compiled in (a separate instantiation: the extra state it threads through the
// visit-map stream costs the plain step 20 % when it is only a run-time flag)
#ifndef LMAZE_WIN_SUB
#define LMAZE_WIN_SUB 64    // envs whose window rows one pass of phase 2 holds in registers
#endif
#ifdef LMAZE_FOVEAL_WAVES   // experiment builds only (tools/_exp): force a register budget
#define LMAZE_FOVEAL_ATTR __attribute__((amdgpu_waves_per_eu(LMAZE_FOVEAL_WAVES)))
#else
#define LMAZE_FOVEAL_ATTR
#endif
template <int VARIANT, int MODE, int EPB, int GT, bool AR>
__global__ __launch_bounds__(LMAZE_BLOCK) LMAZE_FOVEAL_ATTR void foveal_kernel(const FovealArgs a) {
    constexpr bool V1 = VARIANT == LMAZE_VARIANT_V1, V5 = VARIANT == LMAZE_VARIANT_V5;
    constexpr bool V4 = VARIANT == LMAZE_VARIANT_V4 || V5;   // "has a visit map"
    constexpr int C = V1 ? 4 : (V4 ? 7 : 5);
    // reset placement: row reads in flight per pass (mask_counts).  The two-level variants sit at the 128-VGPR step
    // (4 waves per SIMD) and any unrolling there costs a wave; v2/v4 have the room
    constexpr int PLACE_U = V5 ? 1 : 4;
    constexpr int PERENV = C * W25;  // floats of observation per env
    const int G = GT ? GT : a.p.grid, CELLS = G * G, L = V1 ? 1 : a.p.n_layouts;

    // LDS: per-env plane masks and window centres, the two 5x5 samples of the visit map (v4-v6), one
    // 64-bit row mask per layout row for each static plane, and the layout characters for the transition
    extern __shared__ int4 lds4[];
    // the 0/1 planes as bit strings, bit f = float f of the workgroup's contiguous output range (the float visit
    // planes of v4-v6 are zero bits there and come from vwin): a 16-byte store is one nibble of the string
    uint32_t* obits = reinterpret_cast<uint32_t*>(lds4);                   // [EPB*PERENV bits]  obs
    uint32_t* lbits = obits + EPB * 8;                                     // [EPB*100 bits]     obs_local (v5/v6)
    int16_t* cen = reinterpret_cast<int16_t*>(lbits + EPB * 4);            // [EPB][4]  cx, cy, px, py
    int32_t* flags = reinterpret_cast<int32_t*>(cen + EPB * 4);            // [EPB]     bit0 skip, bit1 visit update, bit2 fresh episode, bit3 not stepped
    int16_t* rcen = reinterpret_cast<int16_t*>(flags + EPB);               // [EPB][2]  ball a fused reset placed (visit map re-init)
    float* vwin = reinterpret_cast<float*>(rcen + EPB * 2);                // [EPB][2][25] visit-map samples (v4-v6)
    int32_t* clk = reinterpret_cast<int32_t*>(vwin + (V4 ? EPB * 2 * W25 : 0));   // [EPB] visit clock on entry (v4-v6)
    int32_t* dlist = clk + (V4 ? EPB : 0);                                 // [EPB] envs whose whole map is rewritten this call
    uint64_t* rowfree = reinterpret_cast<uint64_t*>(dlist + (V4 ? EPB : 0));   // [L*G] free = B|S|X
    uint64_t* rowgoal = rowfree + L * G;                                   // [L*G] interior, not 'W', not 'S' (v2:279)
    uint64_t* rowball = rowgoal + L * G;                                   // [L*G] interior, not 'W', not 'X' (v2:292)
    uint64_t* rowwall = rowball + L * G;                                   // [G] v1: 'W'
    uint64_t* rowx = rowwall + G;                                          // [G] v1: 'X'
    uint8_t* lays = reinterpret_cast<uint8_t*>(rowx + G);                  // [L*CELLS]
    static_assert(PERENV <= 8 * 32 - 32 && 4 * W25 <= 4 * 32 - 4, "bit strings fit the 32 B / 16 B per env reserved for them");
    __shared__ int any_skip, ndense;

    const int tid = threadIdx.x;
    // A workgroup takes chunks of EPB envs grid-stride (chunk = blockIdx.x, + gridDim.x, ...; one chunk each unless the
    // launcher asked for more): the set-up below -- row masks of all L layouts -- is paid once per workgroup while the
    // private range it streams at any moment stays one small chunk (lmaze_step.hip step_shared_kernel does the same)
    const int64_t nchunks = (a.n + EPB - 1) / EPB;
    int64_t chunk = blockIdx.x;
    int64_t blockbase = chunk * EPB;
    int nb = (int)min((int64_t)EPB, a.n - blockbase);
    if (tid == 0) { any_skip = 0; ndense = 0; }
    for (int i = tid; i < EPB * 12; i += LMAZE_BLOCK) obits[i] = 0u;       // obits and lbits
    if (V4) for (int i = tid; i < EPB * 2 * W25; i += LMAZE_BLOCK) vwin[i] = 0.0f;   // window cells outside the array read 0
    // the reset epoch, read in front of every store (one uniform scalar load; lmaze_step.hip step_shared_kernel)
    const uint64_t epoch = launch_epoch(a.epoch, a.epoch_in);
    if (MODE == FM_STEP && AR) pass_epoch_on(a.epoch_in, a.epoch_out);
    // large batches: the first 256 workgroups touch every 64-byte line of this step's action array at kernel
    // start, one burst of reads, so that the per-workgroup loads later in the launch hit the memory-side cache
    // instead of turning the saturated write stream around (lmaze_step.hip, step_shared_kernel)
    int warmed = 0;
    if (MODE == FM_STEP && a.nt) {
        warmed = warm_lines(a.action, a.n * 4, 256);
        // v1, v2: the per-env state as well (v2 -3 %).  v4 since its visit map is window-only (round 3: 334-338 us against
        // 341-377 without, three interleaved passes); v5/v6: no gain (418-443 against 425-465) -- experiment switch only
        if (!V5 || LMAZE_WARM_V4(a)) {
            warmed += warm_lines(a.b.ball_xy, a.n * 8, 256) + warm_lines(a.b.step_count, a.n * 4, 256);
            if (V1) warmed += warm_lines(a.b.fgoal_xy, a.n * 8, 256) + warm_lines(a.b.foveal_step_count, a.n * 4, 256);
            else warmed += warm_lines(a.b.goal_xy, a.n * 8, 256) + warm_lines(a.b.layout_id, a.n * 4, 256);
            if (V4) warmed += warm_lines(a.b.visit_clock, a.n * 4, 256);
            if (V5) warmed += warm_lines(a.b.fgoal_xy, a.n * 8, 256) + warm_lines(a.b.foveal_step_count, a.n * 4, 256) +
                              warm_lines(a.b.fovea_xy, a.n * 16, 256) + warm_lines(a.b.ball1_xy, a.n * 8, 256) +
                              warm_lines(a.b.last_xy, a.n * 8, 256) + warm_lines(a.b.foveal_goal, a.n * 4, 256) +
                              warm_lines(a.goal2, a.n * 4, 256);
        }
    }
    if (LMAZE_XP(a, 1)) {
        // experiment: no set-up
    } else if (GT != 0) {
        // Row masks by ballot, straight from global memory: a wave-iteration covers RPW whole layout rows (their
        // characters are RPW*G contiguous bytes, one per lane), four ballots give the rows' masks, and every load of
        // the workgroup -- these and the copy of the characters the transition looks cells up in -- is in flight
        // before the first is used.  One barrier.  (Round 1 copied the characters to LDS byte by byte, barrier, then
        // 90 lanes walked 18 LDS bytes each, twice: 58 of the 520 us of a v5 launch.)
        constexpr int GG = GT ? GT : 1, RPW = 64 / GG, UNR = 8;
        const int wave = tid >> 6, lane = tid & 63, rows = L * G;
        const int rsub = lane / GG, y = lane - rsub * GG;
        const bool dwords = ((reinterpret_cast<uintptr_t>(a.layouts) | (uintptr_t)(L * CELLS)) & 3) == 0;
        uint32_t cw[2] = {0u, 0u};
        if (dwords) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + j * LMAZE_BLOCK < (L * CELLS) >> 2) cw[j] = reinterpret_cast<const uint32_t*>(a.layouts)[tid + j * LMAZE_BLOCK];
        }
        for (int r00 = 0; r00 < rows; r00 += UNR * 4 * RPW) {
            uint8_t cc[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = r00 + (u * 4 + wave) * RPW + rsub;
                cc[u] = (rsub < RPW && row < rows) ? a.layouts[(size_t)row * G + y] : (uint8_t)0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = r00 + (u * 4 + wave) * RPW + rsub;
                const bool in = rsub < RPW && row < rows;
                const uint8_t c = cc[u];
                const unsigned long long bf = __ballot(in && (c == 'B' || c == 'S' || c == 'X'));   // v1:78, v2:94
                const unsigned long long bw = __ballot(in && c == 'W');                              // v1:70
                const unsigned long long bx = __ballot(in && c == 'X');                              // v1:74
                const unsigned long long bs = __ballot(in && c == 'S');
                if (in && y == 0) {
                    const int sh = rsub * GG;
                    const uint64_t keep = GG == 64 ? ~0ull : ((1ull << GG) - 1ull);
                    const uint64_t fr = (bf >> sh) & keep, wl = (bw >> sh) & keep, xx = (bx >> sh) & keep, ss = (bs >> sh) & keep;
                    rowfree[row] = fr;
                    if (V1) { rowwall[row] = wl; rowx[row] = xx; }
                    if (!V1 && (MODE == FM_RESET || (MODE == FM_STEP && AR))) {
                        const int x = row % G;
                        const uint64_t interior = (x >= 1 && x <= G - 2) ? (((1ull << (G - 2)) - 1ull) << 1) : 0ull;
                        rowgoal[row] = fr & ~ss & interior;      // B or X
                        rowball[row] = fr & ~xx & interior;      // B or S
                    }
                }
            }
        }
        if (dwords) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + j * LMAZE_BLOCK < (L * CELLS) >> 2) reinterpret_cast<uint32_t*>(lays)[tid + j * LMAZE_BLOCK] = cw[j];
            for (int i = tid + 2 * LMAZE_BLOCK; i < (L * CELLS) >> 2; i += LMAZE_BLOCK)     // more than 2 KiB of layouts
                reinterpret_cast<uint32_t*>(lays)[i] = reinterpret_cast<const uint32_t*>(a.layouts)[i];
        } else {
            for (int i = tid; i < L * CELLS; i += LMAZE_BLOCK) lays[i] = a.layouts[i];
        }
    } else {
    for (int i = tid; i < L * CELLS; i += LMAZE_BLOCK) lays[i] = a.layouts[i];
    __syncthreads();
    for (int i = tid; i < L * G; i += LMAZE_BLOCK) {
        uint64_t fr = 0, wl = 0, xx = 0;
        for (int y = 0; y < G; ++y) {
            const uint8_t c = lays[i * G + y];
            fr |= (uint64_t)(c == 'B' || c == 'S' || c == 'X') << y;       // v1:78, v2:94
            wl |= (uint64_t)(c == 'W') << y;                                // v1:70
            xx |= (uint64_t)(c == 'X') << y;                                // v1:74
        }
        rowfree[i] = fr;
        if (V1) { rowwall[i] = wl; rowx[i] = xx; }
        if (!V1 && (MODE == FM_RESET || (MODE == FM_STEP && AR))) {
            uint64_t ss = 0;
            for (int y = 0; y < G; ++y) ss |= (uint64_t)(lays[i * G + y] == 'S') << y;
            const int x = i % G;
            const uint64_t interior = (x >= 1 && x <= G - 2) ? (((1ull << (G - 2)) - 1ull) << 1) : 0ull;
            rowgoal[i] = fr & ~ss & interior;      // B or X
            rowball[i] = fr & ~xx & interior;      // B or S
        }
    }
    }
    __syncthreads();

  for (;;) {
    // ---------------- phase 1: one lane per env ----------------
    for (int le = tid; le < (LMAZE_XP(a, 8) ? 0 : nb); le += LMAZE_BLOCK) {
        const int64_t e = blockbase + le;
        EnvRec r;
        r.skip = 0; r.flat = -1; r.action = -1; r.lid = 0; r.gx = r.gy = -9;
        r.b0x = r.b0y = r.b1x = r.b1y = r.f1x = r.f1y = 0; r.upd = 0; r.pad = 0;
        bool fresh = false, nostep = false;   // fused reset: new episode this call / its step was refused
        bool last_is_cur = true;              // v5/v6: the window shown as "previous" from now on is this call's current one
        int rx = 0, ry = 0;                   // ball the fused reset placed
        int bx = a.b.ball_xy[2 * e], by = a.b.ball_xy[2 * e + 1];
        // every per-env input of a step is requested HERE, before anything is branched on: a load that sits behind a branch
        // on another loaded value (the done flag of the fused reset, localDone of the two-level step) is a second global
        // round trip in series -- v4's fused reset cost +58...95 us per launch that way (round 3, tools/_ar_study)
        const int act_in = (MODE == FM_STEP) ? a.action[e] : 0;
        const int sc_ld = (MODE == FM_STEP) ? a.b.step_count[e] : 0;
        const int done_in = (MODE == FM_STEP && AR) ? a.b.done[e] : 0;
        const int goal2_in = (MODE == FM_STEP && AR && V5) ? a.goal2[e] : 0;
        const int vword = (V4 && !(V5 && MODE == FM_PLANNER)) ? a.b.visit_clock[e] : 0;
        const int vclock = vword & 0xff;
        r.px = (int16_t)bx; r.py = (int16_t)by;
        if (MODE != FM_STEP && a.mask && !a.mask[e]) r.skip = 1;
        if (V1) {
            int fgx = a.b.fgoal_xy[2 * e], fgy = a.b.fgoal_xy[2 * e + 1];
            r.action = 1;  // local view unless this is a reset
            if (MODE == FM_STEP) {
                int sc_in = sc_ld;
                if (AR && done_in) {                         // fused reset(): v1:82-93
                    for (int c = 0; c < CELLS; ++c)
                        if (lays[c] == 'S') { bx = c / G; by = c % G; break; }
                    sc_in = 0;
                }
                const int act = act_in;
                const int sc = sc_in + 1;                              // v1:117
                const int fsc = a.b.foveal_step_count[e] + 1;          // v1:118
                float fr = -0.0f, rw = -0.0f;                          // v1:120-121
                bool local_done = false;                               // v1:123
                int ox, oy;
                decode_action(act, ox, oy);                            // v1:125-133
                const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
                const uint8_t c = lays[tx * G + ty];
                if (c == 'W') {                                        // v1:135-138
                    rw = a.p.reward_wall; fr = a.p.reward_wall;
                } else if (c == 'B') {                                 // v1:140-163
                    bx = tx; by = ty;
                    rw = a.p.reward_move; fr = a.p.reward_move;
                    if (bx < fgx - 1 || bx > fgx + 2 || by < fgy - 1 || by > fgy + 2) {
                        local_done = true; fr = a.p.reward_wall;
                    } else if (by == fgy && bx == fgx) {
                        local_done = true; fr = a.p.reward_goal;
                    }
                } else if (c == 'X') {                                 // v1:165-183
                    bx = tx; by = ty;
                    rw = a.p.reward_goal;
                    if (by == fgy && bx == fgx) { local_done = true; fr = a.p.reward_goal; }
                    else fr = a.p.reward_move;
                }
                const bool done = (rw == a.p.reward_goal) || (sc == a.p.step_limit);                         // v1:294-304
                const bool fdone = local_done || fr == a.p.reward_goal || fsc == a.p.foveal_step_limit || done;  // v1:308-324
                a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                a.b.step_count[e] = sc; a.b.foveal_step_count[e] = fsc;
                a.b.reward[e] = rw; a.b.foveal_reward[e] = fr;
                a.b.done[e] = done ? 1 : 0; a.b.foveal_done[e] = fdone ? 1 : 0;
            } else if (MODE == FM_RESET && !r.skip) {
                if (a.place) {                                         // v1:82-84: ball = first 'S'
                    for (int c = 0; c < CELLS; ++c)
                        if (lays[c] == 'S') { bx = c / G; by = c % G; break; }
                    a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                }
                a.b.reward[e] = -0.0f; a.b.foveal_reward[e] = -0.0f;   // v1:90-91
                a.b.step_count[e] = 0;                                 // v1:93 (fovealStepCount kept, v1:94)
                a.b.done[e] = 0; a.b.foveal_done[e] = 0;
                r.action = 0;                                          // v1:100 getGlobalView
            } else if (MODE == FM_SETGOAL && !r.skip) {                // v1:104-110
                fgx = bx + a.action[2 * e] - 2;
                fgy = by + a.action[2 * e + 1] - 2;
                a.b.fgoal_xy[2 * e] = fgx; a.b.fgoal_xy[2 * e + 1] = fgy;
                a.b.foveal_step_count[e] = 0;
            }
            int flat = fgx * G + fgy;                                  // v1:244-245, numpy negative-index wrap
            if (flat < 0) flat += CELLS;
            r.flat = (flat >= 0 && flat < CELLS) ? flat : -1;
        } else if (V5) {
            int lid = clampi(a.b.layout_id[e], 0, L - 1);
            int gx = a.b.goal_xy[2 * e], gy = a.b.goal_xy[2 * e + 1];
            int fg = a.b.foveal_goal[e];
            int f0x = a.b.fovea_xy[4 * e], f0y = a.b.fovea_xy[4 * e + 1];
            int f1x = a.b.fovea_xy[4 * e + 2], f1y = a.b.fovea_xy[4 * e + 3];
            int b1x = a.b.ball1_xy[2 * e], b1y = a.b.ball1_xy[2 * e + 1];
            int lx = a.b.last_xy[2 * e], ly = a.b.last_xy[2 * e + 1];
            if (MODE == FM_STEP) {                                     // v5:187-292
                const int act = act_in;
                int fgx = a.b.fgoal_xy[2 * e], fgy = a.b.fgoal_xy[2 * e + 1];
                int fsc = a.b.foveal_step_count[e];
                int sc_in = sc_ld;
                bool ld = a.b.foveal_done[e] != 0, gd = a.b.done[e] != 0;   // both persist across step() calls
                if (AR) {
                    // the two-level loop around step() (lmaze_v5_hier_step): reset() for an env that enters with
                    // globalDone, plannerStep(goal) for one that enters with localDone or was just reset
                    const bool plan = ld || gd;
                    bool planned = false;
                    if (gd) {                                          // reset(): v5:104-150, as FM_RESET below
                        const uint4 d = env_draw(a.seed, epoch, a.env_base + e);
                        lid = (int)__umulhi(d.z, (uint32_t)L);         // v5:105 setGrid first
                        int goal_cell, ball_cell;
                        place_goal_ball<PLACE_U>(rowgoal + lid * G, rowball + lid * G, G, d, goal_cell, ball_cell);
                        if (goal_cell >= 0) { gx = goal_cell / G; gy = goal_cell % G; }
                        if (ball_cell >= 0) { bx = ball_cell / G; by = ball_cell % G; }
                        a.b.layout_id[e] = lid;
                        a.b.goal_xy[2 * e] = gx; a.b.goal_xy[2 * e + 1] = gy;
                        fsc = 0; sc_in = 0; gd = false; ld = false;    // v5:109-112
                        fg = 12;                                       // v5:127-128
                        f0x = f1x = b1x = lx = fgx = bx; f0y = f1y = b1y = ly = fgy = by;   // v5:136-143
                        fresh = true;
                    }
                    if (plan) {                                        // plannerStep(goal): v5:158-182, as FM_PLANNER below
                        const int g = goal2_in;
                        if (g >= 0 && g < W25) {
                            fg = g;
                            sc_in = 0;                                 // v5:160
                            ld = false;                                // v5:162
                            fgx = bx + g / FOV - 2; fgy = by + g % FOV - 2;   // v5:172-173
                            if (fsc > 0) { f1x = f0x; f1y = f0y; }     // v5:175-177
                            fsc += 1;                                  // v5:179
                            planned = true;
                        }
                    }
                    if (fresh || planned) {
                        a.b.foveal_goal[e] = fg;
                        a.b.fgoal_xy[2 * e] = fgx; a.b.fgoal_xy[2 * e + 1] = fgy;
                        a.b.fovea_xy[4 * e + 2] = f1x; a.b.fovea_xy[4 * e + 3] = f1y;
                        a.b.foveal_step_count[e] = fsc;
                    }
                }
                const uint8_t* lay = lays + lid * CELLS;
                b1x = bx; b1y = by;                                    // v5:193-194
                float lr = -0.0f, gr;                                  // v5:196
                const int sc = sc_in + 1;                              // v5:197
                const int dx = (act == 0) - (act == 1), dy = (act == 2) - (act == 3);   // v5:205-217
                const int nx = bx + dx, ny = by + dy;
                const bool nin = nx >= 0 && ny >= 0 && nx < G && ny < G;
                const uint8_t c = nin ? lay[nx * G + ny] : (uint8_t)'W';
                if (c == 'W') {                                        // v5:232-233
                    lr = a.p.reward_wall;
                } else if (nx == fgx && ny == fgy) {                   // v5:235-239
                    lr = a.p.reward_goal; bx = nx; by = ny; ld = true;
                } else if (c == 'B' || c == 'S' || c == 'X') {         // v5:241-248
                    if (nx < f1x - 3 || nx > f1x + 2 || ny < f1y - 3 || ny > f1y + 2) ld = true;
                    lr = a.p.reward_move; bx = nx; by = ny;
                }
                if (nx == gx && ny == gy) { gr = a.p.reward_goal; gd = true; }   // v5:254-262
                else if (nx == fgx && ny == fgy) gr = a.p.reward_move;
                else gr = a.p.reward_wall;
                f0x = bx; f0y = by;                                    // v5:264-265
                if (sc >= a.p.step_limit) ld = true;                   // v5:267
                if (fsc >= a.p.foveal_step_limit) { gd = true; ld = true; }   // v5:269-271
                if (fsc == 0) { lx = f0x; ly = f0y; }                  // v5:322-323
                r.px = (int16_t)lx; r.py = (int16_t)ly;                // window the foveal obs shows as "previous"
                r.upd = ld ? 1 : 0;                                    // v5:313-318
                if (ld) { lx = f0x; ly = f0y; }                        // v5:344-346 (after the render)
                last_is_cur = lx == f0x && ly == f0y;
                a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                a.b.ball1_xy[2 * e] = b1x; a.b.ball1_xy[2 * e + 1] = b1y;
                a.b.fovea_xy[4 * e] = f0x; a.b.fovea_xy[4 * e + 1] = f0y;
                a.b.last_xy[2 * e] = lx; a.b.last_xy[2 * e + 1] = ly;
                a.b.step_count[e] = sc;
                a.b.reward[e] = gr; a.b.foveal_reward[e] = lr;
                a.b.done[e] = gd ? 1 : 0; a.b.foveal_done[e] = ld ? 1 : 0;
            } else if (MODE == FM_PLANNER && !r.skip) {                // v5:158-182
                const int g = a.action[e];
                if (g < 0 || g >= W25) {
                    r.skip = 1;                                        // the reference raises half-way (v5:169)
                } else {
                    fg = g;
                    a.b.step_count[e] = 0;                             // v5:160
                    a.b.reward[e] = -0.0f;                             // v5:161
                    a.b.foveal_done[e] = 0;                            // v5:162
                    a.b.foveal_goal[e] = g;
                    a.b.fgoal_xy[2 * e] = bx + g / FOV - 2;            // v5:172-173
                    a.b.fgoal_xy[2 * e + 1] = by + g % FOV - 2;
                    const int fsc = a.b.foveal_step_count[e];
                    if (fsc > 0) {                                     // v5:175-177
                        f1x = f0x; f1y = f0y;
                        a.b.fovea_xy[4 * e + 2] = f1x; a.b.fovea_xy[4 * e + 3] = f1y;
                    }
                    a.b.foveal_step_count[e] = fsc + 1;                // v5:179
                }
            } else if (MODE == FM_RESET && !r.skip) {                  // v5:104-150
                if (a.place) {
                    const uint4 d = env_draw(a.seed, a.epoch, a.env_base + e);
                    lid = (int)__umulhi(d.z, (uint32_t)L);             // v5:105 setGrid first
                    a.b.layout_id[e] = lid;
                    int goal_cell, ball_cell;
                    place_goal_ball<PLACE_U>(rowgoal + lid * G, rowball + lid * G, G, d, goal_cell, ball_cell);
                    if (goal_cell >= 0) {
                        gx = goal_cell / G; gy = goal_cell % G;
                        a.b.goal_xy[2 * e] = gx; a.b.goal_xy[2 * e + 1] = gy;
                    }
                    if (ball_cell >= 0) {
                        bx = ball_cell / G; by = ball_cell % G;
                        a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                    }
                }
                a.b.foveal_reward[e] = -0.0f; a.b.reward[e] = -0.0f;   // v5:107-108
                a.b.foveal_step_count[e] = 0; a.b.step_count[e] = 0;   // v5:109-110
                a.b.done[e] = 0; a.b.foveal_done[e] = 0;               // v5:111-112
                fg = 12;                                               // v5:127-128
                a.b.foveal_goal[e] = fg;
                f0x = f1x = b1x = lx = bx; f0y = f1y = b1y = ly = by;  // v5:136-143
                a.b.fovea_xy[4 * e] = bx; a.b.fovea_xy[4 * e + 1] = by;
                a.b.fovea_xy[4 * e + 2] = bx; a.b.fovea_xy[4 * e + 3] = by;
                a.b.fgoal_xy[2 * e] = bx; a.b.fgoal_xy[2 * e + 1] = by;
                a.b.ball1_xy[2 * e] = bx; a.b.ball1_xy[2 * e + 1] = by;
                a.b.last_xy[2 * e] = bx; a.b.last_xy[2 * e + 1] = by;
                r.px = (int16_t)bx; r.py = (int16_t)by;
            }
            r.lid = (int16_t)lid;
            r.gx = (int16_t)gx; r.gy = (int16_t)gy;
            r.action = (int16_t)fg;
            r.b0x = (int16_t)bx; r.b0y = (int16_t)by; r.b1x = (int16_t)b1x; r.b1y = (int16_t)b1y;
            r.f1x = (int16_t)f1x; r.f1y = (int16_t)f1y;
            bx = f0x; by = f0y;                                        // the window centre is fovea_0
        } else {
            int lid = a.b.layout_id[e];
            int gx = a.b.goal_xy[2 * e], gy = a.b.goal_xy[2 * e + 1];
            int sc_in = 0;
            const bool fused = MODE == FM_STEP && AR && done_in != 0;
            if ((MODE == FM_RESET && !r.skip) || fused) {              // reset(): v2:80-123, v4:95-163
                if (a.place || fused) {
                    const uint4 d = env_draw(a.seed, epoch, a.env_base + e);
                    const int lid_new = (int)__umulhi(d.z, (uint32_t)L);
                    if (V4) lid = lid_new;                             // v4:97 setGrid first
                    lid = clampi(lid, 0, L - 1);
                    int goal_cell, ball_cell;
                    place_goal_ball<PLACE_U>(rowgoal + lid * G, rowball + lid * G, G, d, goal_cell, ball_cell);
                    if (goal_cell >= 0) {
                        gx = goal_cell / G; gy = goal_cell % G;
                        a.b.goal_xy[2 * e] = gx; a.b.goal_xy[2 * e + 1] = gy;
                    }
                    if (ball_cell >= 0) {
                        bx = ball_cell / G; by = ball_cell % G;
                        a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                    }
                    lid = lid_new;                                     // v2:92 setGrid last
                    a.b.layout_id[e] = lid;
                }
                a.b.reward[e] = -0.0f;                                 // v2:84
                a.b.step_count[e] = 0;                                 // v2:86
                a.b.done[e] = 0;
                r.px = (int16_t)bx; r.py = (int16_t)by;                // v2:109: previous = current
                fresh = true;
                rx = bx; ry = by;
            } else if (MODE == FM_STEP) {
                sc_in = sc_ld;
            }
            if (MODE == FM_STEP) {
                const int act = act_in;
                if (act < 0 || act >= W25) {
                    if (fresh) nostep = true;                          // reset, then the reference's step() raises
                    else r.skip = 1;                                   // the reference raises before touching anything
                } else {
                    lid = clampi(lid, 0, L - 1);
                    const uint8_t* lay = lays + lid * CELLS;
                    float rw = -0.0f;                                  // v2:146
                    const int sc = sc_in + 1;                          // v2:147
                    const int fx = bx + act / FOV - 2, fy = by + act % FOV - 2;   // v2:151-152
                    if (fx < G - 2 && fx > 1 && fy < G - 2 && fy > 1) {           // v2:157-159
                        bx = fx; by = fy;
                    } else {                                           // v2:160-169
                        if (fx >= G - 2) bx = G - 3;
                        if (fx <= 1) bx = 2;
                        if (fy >= G - 2) by = G - 3;
                        if (fy <= 1) by = 2;
                    }
                    const bool fin = fx >= 0 && fy >= 0 && fx < G && fy < G;
                    const uint8_t c = fin ? lay[fx * G + fy] : (uint8_t)'W';
                    if (fx == gx && fy == gy) rw = a.p.reward_goal;    // v2:175-180
                    else if (c == 'W') rw = a.p.reward_wall;
                    else if (c == 'B' || c == 'S') rw = a.p.reward_move;
                    a.b.ball_xy[2 * e] = bx; a.b.ball_xy[2 * e + 1] = by;
                    a.b.step_count[e] = sc;
                    a.b.reward[e] = rw;
                    a.b.done[e] = (rw == a.p.reward_goal || sc > a.p.step_limit) ? 1 : 0;   // v2:222
                    r.action = (int16_t)act;
                }
            }
            r.lid = (int16_t)clampi(lid, 0, L - 1);
            r.gx = (int16_t)gx; r.gy = (int16_t)gy;
        }
        r.cx = (int16_t)bx; r.cy = (int16_t)by;
        // the observation as 25-bit planes (the float visit planes are sampled in phase 3)
        uint32_t m[8];
        if (V1) {
            m[0] = 1u << 12;                                                               // ball, v1:216
            m[1] = window_bits(rowwall, G, r.cx, r.cy);
            m[2] = r.action ? (r.flat >= 0 ? onehot_bits(r.flat / G, r.flat % G, r.cx, r.cy) : 0u)   // v1:244-245
                            : window_bits(rowx, G, r.cx, r.cy);
            m[3] = window_bits(rowfree, G, r.cx, r.cy);
        } else {
            constexpr int PER = VARIANT == LMAZE_VARIANT_V2 ? 2 : 3;
            const uint64_t* rows = rowfree + r.lid * G;
            m[0] = window_bits(rows, G, r.cx, r.cy);                                       // v2:94
            m[1] = onehot_bits(r.gx, r.gy, r.cx, r.cy);                                    // v2:95
            m[PER] = (r.action >= 0 && r.action < W25) ? (1u << r.action) : 0u;            // v2:135-136, v5:166-169
            m[PER + 1] = window_bits(rows, G, r.px, r.py);
            m[PER + 2] = onehot_bits(r.gx, r.gy, r.px, r.py);
            if (V5) {                                                                      // v5:356-380
                uint32_t lm[4];
                lm[0] = m[0];
                const int i0 = wrap5(r.b0x - r.f1x + 2), j0 = wrap5(r.b0y - r.f1y + 2);
                const int i1 = wrap5(r.b1x - r.f1x + 2), j1 = wrap5(r.b1y - r.f1y + 2);
                lm[1] = (i0 >= 0 && j0 >= 0) ? (1u << (FOV * i0 + j0)) : 0u;
                lm[2] = (i1 >= 0 && j1 >= 0) ? (1u << (FOV * i1 + j1)) : 0u;
                lm[3] = m[PER];
                if (MODE != FM_RESET && !r.skip)
                    for (int ch = 0; ch < 4; ++ch) put_bits(lbits, le * (4 * W25) + ch * W25, lm[ch]);
            }
        }
        if (!r.skip) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (!(V4 && (ch == 2 || ch == 6))) put_bits(obits, le * PERENV + ch * W25, m[ch]);
        }
        cen[le * 4 + 0] = r.cx; cen[le * 4 + 1] = r.cy; cen[le * 4 + 2] = r.px; cen[le * 4 + 3] = r.py;
        int fl = (r.skip ? 1 : 0) | (r.upd ? 2 : 0) | (fresh ? 4 : 0) | (nostep ? 8 : 0);
        if (V4 && !(V5 && MODE == FM_PLANNER) && !r.skip) {
            // What this call does to the env's visit map, in clock terms (phase 2 carries it out on the cells):
            //   zero    reset(): the map restarts from zeros, clock 0 (v4:112, v5:130)
            //   renorm  the clock is about to leave the exponent range: rewrite the map in true values, clock VISIT_BIAS
            //   pre     fused reset of v4: the reset's own (0 + window) / 2 at the placed ball (v4:116-119)
            //   add     this call's (map + window) / 2 at the window centre: v4 every step and every reset
            //           (v4:211-214), v5/v6 only on localDone (v5:313-318) and never at reset (v5:130)
            const bool zero = MODE == FM_RESET || (MODE == FM_STEP && AR && fresh);
            const bool renorm = !zero && vclock >= VISIT_RENORM;
            const bool pre = !V5 && MODE == FM_STEP && AR && fresh;
            const bool add = !(V5 && MODE == FM_RESET) && !nostep && !(V5 && MODE == FM_STEP && !r.upd);
            // the "previous window" record behind the tiles: it must hold the window the NEXT call shows as previous, in
            // true values -- this call's current window (cw0: v4 always; v5/v6 when retStatelast moved, v5:322-346), or
            // the previous one as this call left it (cw1: the map or the episode changed under it)
            const bool cw0 = V5 && (MODE == FM_RESET || last_is_cur);
            const bool cw1 = V5 && !cw0 && (fresh || add);
            // v5/v6: does the env's record hold the window this call shows as "previous"?  (a freshly loaded state does not)
            const bool hit = V5 && !zero && (vword >> 8) == (visit_tag(r.px, r.py) >> 8);
            fl |= (zero ? 16 : 0) | (renorm ? 32 : 0) | (pre ? 64 : 0) | (add ? 128 : 0) | (cw0 ? 256 : 0) | (cw1 ? 512 : 0) | (hit ? 1024 : 0);
            int c1 = (zero ? 0 : (renorm ? VISIT_BIAS : vclock)) + (pre ? 1 : 0) + (add ? 1 : 0);
            // the record's tag: the centre it will hold after this call (cw0 / cw1), else as it was
            if (V5) c1 |= cw0 ? visit_tag(r.cx, r.cy) : (cw1 ? visit_tag(r.px, r.py) : (vword & ~0xff));
            if (c1 != vword) a.b.visit_clock[e] = c1;
            if (zero || renorm) dlist[atomicAdd(&ndense, 1)] = le;
        }
        if (V4) clk[le] = vclock;
        flags[le] = fl;
        rcen[le * 2] = (int16_t)rx; rcen[le * 2 + 1] = (int16_t)ry;
        if (r.skip) any_skip = 1;
    }
    __syncthreads();
    const bool some_skipped = any_skip != 0;

    // ---------------- phase 2 (v4-v6): the visit maps, v4:116-119 / v4:211-214 / v5:313-318 ----------------
    // Clock-relative tiles (include/lmaze.h "The visit map"): the whole-plane halving already happened in phase 1 (the
    // env's clock moved); what is left is the 5x5 window.  The few envs whose whole map is rewritten (reset: zeros; clock at
    // VISIT_RENORM: true values) are streamed tile by tile; everybody else goes through the window pass below, one lane per
    // window row.  (Tried and dropped this round, LAB_NOTES.md R3.1 / R3.5: one lane per TILE row -- the bookkeeping made the
    // phase issue-bound --, whole tiles staged in LDS by LDS-DMA -- 640 B of LDS per env --, one wave instruction per env, and
    // a software-pipelined chunk loop that issues the next chunk's phase 1 and tile loads before this chunk's stores --
    // 45 registers of loads held across the store phase: 171-214 VGPRs, 2-3 waves per SIMD, 357 / 552 us against 296 / 422.)
    if (V4 && !(V5 && MODE == FM_PLANNER) && !LMAZE_XP(a, 32)) {
        const int TB = visit_tiles(G), TILES = TB * TB;
        uint32_t* vis = reinterpret_cast<uint32_t*>(a.b.visit) + (size_t)blockbase * TILES * (VT * VT);
        uint32_t* rec = reinterpret_cast<uint32_t*>(a.b.visit) + (size_t)a.n * TILES * (VT * VT) + (size_t)blockbase * VPC;
        // One tile row of an env whose WHOLE map is rewritten this call: zeros (reset) or true values (clock at
        // VISIT_RENORM) first, then the fused reset's own window at the placed ball (`pre`, v4:116-119), then `add`.
        auto tile_row_whole = [&](uint32_t (&s)[4], int x, int y0, int le, int fl) {
            const int E0 = clk[le];
            const bool zero = fl & 16, renorm = fl & 32, pre = fl & 64, add = fl & 128;
            const int cx = cen[le * 4], cy = cen[le * 4 + 1], px = cen[le * 4 + 2], py = cen[le * 4 + 3];
            const int E1 = zero ? 0 : (renorm ? VISIT_BIAS : E0);
            const int dx = x - cx + 2, ex = x - px + 2;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int y = y0 + k;
                const bool cell_ok = x < G && y < G;      // tiles are padded up to a multiple of 4: those cells stay 0
                uint32_t b = s[k];
                if (zero) b = 0u;
                else if (renorm) b = visit_true(b, E0);
                int E = E1;
                if (pre) {
                    const int qx = x - rcen[le * 2] + 2, qy = y - rcen[le * 2 + 1] + 2;
                    if (cell_ok && (unsigned)qx <= 4u && (unsigned)qy <= 4u) b = visit_add(b, E);
                    ++E;
                }
                const int dy = y - cy + 2, ey = y - py + 2;
                const bool in = cell_ok && (unsigned)dx <= 4u && (unsigned)dy <= 4u;
                if (add) {
                    if (in) b = visit_add(b, E);
                    ++E;
                }
                s[k] = b;
                if (in) vwin[le * 2 * W25 + dx * FOV + dy] = __uint_as_float(visit_true(b, E));
                if (cell_ok && (unsigned)ex <= 4u && (unsigned)ey <= 4u)
                    vwin[le * 2 * W25 + W25 + ex * FOV + ey] = __uint_as_float(visit_true(b, E));
            }
        };
        // ---- whole maps first: envs that were reset (zeros, nothing loaded) or whose clock reached VISIT_RENORM
        {
            const int nd = ndense, per = TILES * VT;
            for (int j = tid; j < nd * per; j += LMAZE_BLOCK) {
                const int d = j / per, r = j - d * per;
                const int le = dlist[d], fl = flags[le];
                const int tile = r >> 2, row = r & 3;
                const int tx = tile / TB, ty = tile - tx * TB;
                uint32_t* p = vis + (le * TILES + tile) * (VT * VT) + row * VT;
                uint32_t sv[4] = {0u, 0u, 0u, 0u};
                if (!(fl & 16)) {
                    const uint4 t4 = *reinterpret_cast<const uint4*>(p);
                    sv[0] = t4.x; sv[1] = t4.y; sv[2] = t4.z; sv[3] = t4.w;
                }
                tile_row_whole(sv, tx * VT + row, ty * VT, le, fl & 0xff);
                *reinterpret_cast<uint4*>(p) = make_uint4(sv[0], sv[1], sv[2], sv[3]);
            }
        }
        // ---- the windows: ONE LANE PER WINDOW ROW.  Item = (env, window 0: current / 1: "previous", row 0..4): the row's
        // five cells lie in two horizontally adjacent tiles, on one tile row each -- two 16-byte loads, eight words, the
        // five wanted ones start at word y0 & 3 --, or, v5/v6, in the env's "previous window" record (20 contiguous
        // bytes) when that window lies elsewhere.  Every load of the chunk is in flight before the first is used, and
        // nothing is stored before every lane has its loads (the barrier): a cell both windows show is loaded by two
        // lanes and each works out the same new value for it.  Cells of the current window take (v + 1) / 2 when the map
        // updates this call and the two 16-byte pieces go back re-encoded under the new clock (into lines the loads have
        // just brought into L2); the TRUE values both windows show -- the previous one sampled live from the updated
        // map, Appendix B-7 -- are left in vwin for phase 3.
        {
            // SUB envs at a time (one barrier each): the loads of a pass are held in registers, 9 per row
            constexpr int IPE = 2 * FOV, SUB = EPB < LMAZE_WIN_SUB ? EPB : LMAZE_WIN_SUB, NIT = (SUB * IPE + LMAZE_BLOCK - 1) / LMAZE_BLOCK;
          for (int sb = 0; sb < nb; sb += SUB) {
            const int items = LMAZE_XP(a, 256) ? 0 : min(SUB, nb - sb) * IPE;
            uint4 va[NIT], vb[NIT];
            int meta[NIT];    // -1 nothing; else global word offset of piece A (tiles) or of the row (record) | 1 << 28 record | 1 << 29 piece A outside | 1 << 30 piece B outside
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + u * LMAZE_BLOCK;
                meta[u] = -1;
                va[u] = make_uint4(0u, 0u, 0u, 0u);
                vb[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i >= items) continue;
                const int le = sb + i / IPE, r = i % IPE;
                const int w = r >= FOV ? 1 : 0, row = r - w * FOV;
                const int fl = flags[le];
                if (fl & (1 | 16 | 32)) continue;                           // untouched, or rewritten whole above
                const int cx = cen[le * 4], cy = cen[le * 4 + 1];
                const int wx = cen[le * 4 + 2 * w], wy = cen[le * 4 + 2 * w + 1];
                const int x = wx - 2 + row, y0 = wy - 2;
                if ((unsigned)x >= (unsigned)G) continue;                   // outside the array: stays 0
                // v5/v6: the record serves the previous window unless this call's update reaches into this row
                const bool touched = (fl & 128) && (unsigned)(x - cx + 2) <= 4u && (unsigned)(wy - cy + 4) <= 8u;
                if (V5 && w == 1 && (fl & 1024) && !touched) {
                    const int o = le * VPC + row * FOV;
                    struct __attribute__((packed, aligned(4))) Q4 { uint32_t v[4]; };
                    const Q4 q = *reinterpret_cast<const Q4*>(rec + o);
                    va[u] = make_uint4(q.v[0], q.v[1], q.v[2], q.v[3]);
                    vb[u].x = rec[o + 4];
                    meta[u] = o | (1 << 28);
                } else {
                    const int ty0 = y0 >> 2;                                // floor: -1 when the window pokes out on the left
                    const int o = (le * TILES + (x >> 2) * TB + ty0) * (VT * VT) + (x & 3) * VT;
                    const bool a_out = ty0 < 0, b_out = ty0 + 1 >= TB;
                    if (!a_out) va[u] = *reinterpret_cast<const uint4*>(vis + o);
                    if (!b_out) vb[u] = *reinterpret_cast<const uint4*>(vis + o + VT * VT);
                    meta[u] = (o & 0x0fffffff) | (a_out ? 1 << 29 : 0) | (b_out ? 1 << 30 : 0);
                }
            }
            __syncthreads();       // every load of this chunk's envs has returned before any of their cells is stored
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                if (meta[u] < 0) continue;
                const int i = tid + u * LMAZE_BLOCK;
                const int le = sb + i / IPE, r = i % IPE;
                const int w = r >= FOV ? 1 : 0, row = r - w * FOV;
                const int fl = flags[le], E0 = clk[le];
                const bool add = fl & 128, from_rec = (meta[u] >> 28) & 1;
                const int cx = cen[le * 4], cy = cen[le * 4 + 1];
                const int x = cen[le * 4 + 2 * w] - 2 + row, y0 = cen[le * 4 + 2 * w + 1] - 2;
                const int sh = from_rec ? 0 : (y0 & 3);
                // the eight words rotated so that the row's cells are c[0..4]
                uint32_t c[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                if (sh & 1) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) c[k] = c[k + 1];
                }
                if (sh & 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) c[k] = c[k + 2];
                }
                const bool rowc = (unsigned)(x - cx + 2) <= 4u;
                bool changed = false;
                float* out = vwin + le * 2 * W25 + w * W25 + row * FOV;
                const bool rec_out = V5 && (fl & (w == 0 ? 256 : 512));     // v5/v6: the record takes the window the NEXT call shows as previous
                // One cell: its true value under the clock this call ends with (a record holds true values, i.e. values stored
                // under clock VISIT_BIAS, and serves a row only when this call's update does not reach into it; a cell outside
                // the current window only takes the whole-plane halving of this call's update, if there is one).  `slow`: decode
                // by lmaze_visit_true (values that decayed below 2^-126) instead of the exponent subtraction.
                auto one_cell = [&](uint32_t& cj, int j, bool slow) -> bool {
                    const int y = y0 + j;
                    if ((unsigned)y >= (unsigned)G) return false;           // outside the array: stays 0
                    const bool in_cur = !from_rec && rowc && (unsigned)(y - cy + 2) <= 4u;
                    const int E = in_cur ? E0 : (from_rec ? VISIT_BIAS : E0) + (add ? 1 : 0);
                    const int n = E - VISIT_BIAS, f = (int)(cj >> 23);
                    const bool fast = cj == 0u || (f >= 1 && f - n >= 1);
                    uint32_t t = slow ? visit_true(cj, E) : (cj == 0u ? 0u : (uint32_t)((int)cj - n * (1 << 23)));
                    if (in_cur && add) {
                        const float nv = (__uint_as_float(t) + 1.0f) * 0.5f;    // v4:214; see lmaze_visit_add
                        t = __float_as_uint(nv);
                        if (fast || slow) cj = lmaze_visit_store(nv, E0 + 1);
                        changed = true;
                    }
                    out[j] = __uint_as_float(t);
                    if (rec_out) rec[le * VPC + row * FOV + j] = t;
                    return !fast;
                };
                uint32_t redo = 0u;
#pragma unroll
                for (int j = 0; j < FOV; ++j) redo |= one_cell(c[j], j, false) ? 1u << j : 0u;
#pragma unroll 1
                for (; redo; redo &= redo - 1u) {                           // rare: cells below 2^-126
                    const int j = __ffs((int)redo) - 1;
                    uint32_t cj = j == 0 ? c[0] : (j == 1 ? c[1] : (j == 2 ? c[2] : (j == 3 ? c[3] : c[4])));
                    one_cell(cj, j, true);
                    c[0] = j == 0 ? cj : c[0]; c[1] = j == 1 ? cj : c[1]; c[2] = j == 2 ? cj : c[2];
                    c[3] = j == 3 ? cj : c[3]; c[4] = j == 4 ? cj : c[4];
                }
                if (w == 0 && changed) {
                    // the updated words back where they came from: word k of the row sits at c[k - sh] for k >= sh
                    uint32_t d[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
#pragma unroll
                        for (int j = 0; j < FOV; ++j)
                            if (k - j >= 0 && k - j <= 3 && sh == k - j) d[k] = c[j];
                    }
                    const int o = meta[u] & 0x0fffffff;
                    if (!((meta[u] >> 29) & 1)) *reinterpret_cast<uint4*>(vis + o) = make_uint4(d[0], d[1], d[2], d[3]);
                    if (!((meta[u] >> 30) & 1)) *reinterpret_cast<uint4*>(vis + o + VT * VT) = make_uint4(d[4], d[5], d[6], d[7]);
                }
            }
            if (V5) {
                // rows of a new record that lie outside the array hold zeros
                for (int i = tid; i < items; i += LMAZE_BLOCK) {
                    const int le = sb + i / IPE, r = i % IPE;
                    const int w = r >= FOV ? 1 : 0, row = r - w * FOV;
                    const int fl = flags[le];
                    if ((fl & (1 | 16 | 32)) || !(fl & (w == 0 ? 256 : 512))) continue;
                    const int x = cen[le * 4 + 2 * w] - 2 + row, y0 = cen[le * 4 + 2 * w + 1] - 2;
                    for (int j = 0; j < FOV; ++j)
                        if ((unsigned)x >= (unsigned)G || (unsigned)(y0 + j) >= (unsigned)G) rec[le * VPC + row * FOV + j] = 0u;
                }
            }
          }
        }
        __syncthreads();
        if (V5) {
            // whole-map envs (reset / renormalised this call): their window values are in vwin now
            const int nd = ndense;
            for (int j = tid; j < nd * W25; j += LMAZE_BLOCK) {
                const int le = dlist[j / W25], k = j % W25, fl = flags[le];
                if (!(fl & (256 | 512))) continue;
                const int wsel = (fl & 256) ? 0 : 1;
                rec[le * VPC + k] = __float_as_uint(vwin[le * 2 * W25 + wsel * W25 + k]);
            }
        }
    }

    // ---------------- phase 3: render float[nb*C*25], contiguous, 16-byte stores ----------------
    // float `rem` of env le's observation: a bit of the string, or -- visit planes 2 and 6 of v4-v6, sampled live at
    // the current / "previous" window -- one of the env's 2 x 25 samples
    auto element = [&](int le, int rem) -> float {
        if (V4 && rem >= 2 * W25 && rem < 3 * W25) return vwin[le * 2 * W25 + rem - 2 * W25];
        if (V4 && rem >= 6 * W25) return vwin[le * 2 * W25 + rem - 5 * W25];
        const int f = le * PERENV + rem;
        return ((obits[f >> 5] >> (f & 31)) & 1u) ? 1.0f : 0.0f;
    };
    float* obs = a.b.obs + (size_t)blockbase * PERENV;
    const int R = (V5 && MODE == FM_PLANNER) ? 0 : nb * PERENV;   // plannerStep returns only the local observation
    const int nq = some_skipped ? 0 : (R >> 2);
    if (LMAZE_XP(a, 16) && !V4 && EPB == 128) {
        // experiment: the 4-KiB pieces of 8 consecutive workgroups interleaved (piece k*8 + w of the group's 1024
        // envs), content from this workgroup's own bit string (garbage addresses-wise)
        const int w = blockIdx.x & 7;
        float* gbase = a.b.obs + (size_t)(blockIdx.x >> 3) * 1024 * PERENV;
        const int npieces = 1024 * PERENV / 1024;
        for (int k = 0; k * 8 + w < npieces; ++k) {
            float v[4];
            nibble_floats(obits, (k * 256 + tid) % (EPB * PERENV / 4), v);
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t = {v[0], v[1], v[2], v[3]};
            v4f* dst = reinterpret_cast<v4f*>(gbase) + (size_t)(k * 8 + w) * 256 + tid;
            if (a.nt) stream_store16(dst, t); else *dst = t;
        }
    }
    const int nq_run = (LMAZE_XP(a, 4) || (LMAZE_XP(a, 16) && !V4 && EPB == 128)) ? 0 : nq;
    for (int q = tid; q < nq_run; q += LMAZE_BLOCK) {
        const int f = q << 2;
        int le = f / PERENV;
        int rem = f - le * PERENV;
        float v[4];
        if (!V4) {
            // bit planes only (v1, v2): float f of the workgroup's range is bit f of the string phase 1 left in LDS,
            // a 16-byte store is nibble q of it -- a dozen VALU instructions per store, no index arithmetic
            // (masks per plane and a division per float made this loop ALU-bound: 1 210 VALU per wave on v2)
            nibble_floats(obits, q, v);
        } else {
            // v4-v6: five 0/1 planes and two float planes per env -- the nibble as above, then the floats that fall
            // into a visit plane are replaced by their samples (2 of 7 planes; a third of the stores touch one)
            nibble_floats(obits, q, v);
            if (rem + 3 >= 2 * W25 && !(rem >= 3 * W25 && rem + 3 < 6 * W25)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = rem + k, l2 = le;
                    if (r >= PERENV) { r -= PERENV; ++l2; }
                    if (r >= 2 * W25 && r < 3 * W25) v[k] = vwin[l2 * 2 * W25 + r - 2 * W25];
                    else if (r >= 6 * W25) v[k] = vwin[l2 * 2 * W25 + r - 5 * W25];
                }
            }
        }
        if (a.nt) {  // large batches: the observation cannot stay in the Infinity Cache, stream it (+6...12 %)
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t = {v[0], v[1], v[2], v[3]};
            stream_store16(reinterpret_cast<v4f*>(obs) + q, t);
        } else {
            reinterpret_cast<float4*>(obs)[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    // scalar path: the ragged tail, or every element when some env of the workgroup is skipped
    for (int f = (nq << 2) + tid; f < R; f += LMAZE_BLOCK) {
        const int le = f / PERENV;
        if (flags[le] & 1) continue;
        obs[f] = element(le, f - le * PERENV);
    }

    // ---------------- phase 3b (v5/v6): the local observation float[nb*4*25], v5:356-380 ----------------
    if (V5 && MODE != FM_RESET && !LMAZE_XP(a, 4)) {
        constexpr int PERLOC = 4 * W25;
        float* loc = a.b.obs_local + (size_t)blockbase * PERLOC;
        const int RL = nb * PERLOC;                      // 100 floats per env: a store never straddles two envs
        for (int q = tid; q < (RL >> 2); q += LMAZE_BLOCK) {
            const int f = q << 2;
            const int le = f / PERLOC;
            if (flags[le] & 1) continue;
            float v[4];
            nibble_floats(lbits, q, v);
            if (a.nt) {   // streamed like the foveal observation (round 3: these 400 B per env were plain stores)
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f t = {v[0], v[1], v[2], v[3]};
                stream_store16(reinterpret_cast<v4f*>(loc) + q, t);
            } else {
                reinterpret_cast<float4*>(loc)[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    chunk += gridDim.x;
    if (chunk >= nchunks) break;                                           // uniform over the workgroup
    blockbase = chunk * EPB;
    nb = (int)min((int64_t)EPB, a.n - blockbase);
    __syncthreads();                                                       // every wave is done with this chunk's strings and flags
    for (int i = tid; i < EPB * 12; i += LMAZE_BLOCK) obits[i] = 0u;
    if (V4) for (int i = tid; i < EPB * 2 * W25; i += LMAZE_BLOCK) vwin[i] = 0.0f;
    if (tid == 0) { any_skip = 0; ndense = 0; }
    __syncthreads();
  }
    if (warmed == 0x7fedcba9 && a.n < 0) a.b.done[0] = 1;   // never true: keeps the warming loads alive
}

// v6 safeFovealGoal (v6:505-523): one lane per env
__global__ __launch_bounds__(LMAZE_BLOCK) void safe_goal_kernel(const FovealArgs a, int32_t* out_goal) {
    const int64_t e = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x;
    if (e >= a.n) return;
    const int G = a.p.grid;
    const uint8_t* lay = a.layouts + (size_t)clampi(a.b.layout_id[e], 0, a.p.n_layouts - 1) * G * G;
    const int bx = a.b.ball_xy[2 * e], by = a.b.ball_xy[2 * e + 1];
    const uint4 d = env_draw(a.seed, a.epoch, a.env_base + e);
    unsigned ok = 0;
    for (int c = 0; c < W25; ++c) {
        const int x = bx - 2 + c / FOV, y = by - 2 + c % FOV;
        const bool in = x >= 0 && y >= 0 && x < G && y < G;
        if (in && lay[x * G + y] != 'W') ok |= 1u << c;
    }
    int pick = 12;
    const int cnt = __popc(ok);
    if (cnt > 0) {
        int k = (int)__umulhi(d.x, (uint32_t)cnt);
        for (int c = 0; c < W25; ++c)
            if (ok & (1u << c)) {
                if (k == 0) { pick = c; break; }
                --k;
            }
    }
    out_goal[e] = pick;
}

// ------------------------------------------------------------------------------------
// xE nearest-neighbour on float planes (v1:258-277, v2:197-203): one workgroup per env
// ------------------------------------------------------------------------------------
struct ExpandPlanesArgs {
    const float* planes;
    float* out;
    int64_t n;
    int32_t channels, g, expansion;
};

__global__ __launch_bounds__(LMAZE_BLOCK) void expand_planes_kernel(const ExpandPlanesArgs a) {
    extern __shared__ int4 lds4[];
    const int g = a.g, E = a.expansion, C = a.channels;
    const int PC = g * g, S = g * E, PLANE = S * S, L = C * PLANE;
    float* src = reinterpret_cast<float*>(lds4);                    // [C*g*g]
    uint16_t* rowmap = reinterpret_cast<uint16_t*>(src + C * PC);   // [S] row -> (row / E) * g
    uint16_t* colmap = rowmap + S;                                  // [S] col -> col / E
    const int tid = threadIdx.x;
    for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        __syncthreads();
        for (int k = tid; k < C * PC; k += LMAZE_BLOCK) src[k] = a.planes[(size_t)i * C * PC + k];
        for (int k = tid; k < S; k += LMAZE_BLOCK) {
            rowmap[k] = (uint16_t)((k / E) * g);
            colmap[k] = (uint16_t)(k / E);
        }
        __syncthreads();
        const size_t B = (size_t)i * L;
        const size_t a0 = (B + 3) & ~(size_t)3, a1 = (B + L) & ~(size_t)3;
        auto value = [&](int local) -> float {
            const int c = local / PLANE;
            const int rem = local - c * PLANE;
            const int row = rem / S, col = rem - row * S;
            return src[c * PC + rowmap[row] + colmap[col]];
        };
        if (tid < (int)(a0 - B)) a.out[B + tid] = value(tid);
        if (tid < (int)(B + L - a1)) a.out[a1 + tid] = value((int)(a1 - B) + tid);
        const int nq = (int)((a1 - a0) >> 2);
        float4* out4 = reinterpret_cast<float4*>(a.out + a0);
        const int head = (int)(a0 - B);
        for (int q = tid; q < nq; q += LMAZE_BLOCK) {
            const int local = head + (q << 2);
            int c = local / PLANE;
            int rem = local - c * PLANE;
            int row = rem / S, col = rem - row * S;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = src[c * PC + rowmap[row] + colmap[col]];
                if (++col == S) {
                    col = 0;
                    if (++row == S) { row = 0; ++c; }
                }
            }
            out4[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// The foveal shape itself (5x5 window, x7; every foveal variant): g and E at compile time, so the flat index
// decodes by multiplication, and the output treated as what it is -- ONE contiguous stream of N*C planes of
// 35x35 floats.  Workgroup w writes the aligned stretch [w*CH, (w+1)*CH) floats of it (an env is 19.6-34 KB
// and starts on a 16-byte boundary only: per-env workgroups straddle cache lines with every wave store).
template <int GT, int ET, bool NT>
__global__ __launch_bounds__(LMAZE_BLOCK) void expand_planes_stream_kernel(const ExpandPlanesArgs a, int chunk_floats) {
    constexpr int PC = GT * GT, S = GT * ET, PLANE = S * S;
    extern __shared__ int4 lds4[];
    float* src = reinterpret_cast<float*>(lds4);                    // [planes touched][PC]
    const int tid = threadIdx.x;
    const int64_t total = a.n * (int64_t)a.channels * PLANE;
    const int64_t f0 = (int64_t)blockIdx.x * chunk_floats;
    const int len = (int)min((int64_t)chunk_floats, total - f0);
    const int64_t p0 = f0 / PLANE;                                  // first plane of the stretch (plane = env*C + c)
    const int off0 = (int)(f0 - p0 * PLANE);
    const int np = (off0 + len + PLANE - 1) / PLANE;
    for (int k = tid; k < np * PC; k += LMAZE_BLOCK) src[k] = a.planes[(size_t)p0 * PC + k];
    __syncthreads();
    float* dst = a.out + f0;
    static_assert(ET >= 4, "four consecutive output columns span at most two cells");
    for (int q = tid; (q << 2) < len; q += LMAZE_BLOCK) {
        const int local = off0 + (q << 2);
        const int p = local / PLANE;
        const int rem = local - p * PLANE;
        const int row = rem / S, col = rem - row * S;
        // one path for every lane (S = 35: every wave holds float4s that straddle an output row): the value
        // under the first column, the next one of the same window row, and the first of the following row
        const float* r0 = src + p * PC + (row / ET) * GT;
        const int k0 = col / ET;
        const float v0 = r0[k0], v1 = r0[k0 + 1];                      // r0[GT] is read but never selected
        int row1 = row + 1, p1 = p;
        if (row1 == S) { row1 = 0; ++p1; }
        const float vw = src[p1 * PC + (row1 / ET) * GT];
        const int edge = (k0 + 1) * ET;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = col + j;
            v[j] = cj >= S ? vw : (cj >= edge ? v1 : v0);
        }
        const int count = len - (q << 2);
        if (count >= 4) {
            if (NT) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 t = {v[0], v[1], v[2], v[3]};
                stream_store16(reinterpret_cast<f4*>(dst) + q, t);
            } else {
                reinterpret_cast<float4*>(dst)[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
            for (int j = 0; j < count; ++j) dst[(q << 2) + j] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
constexpr size_t kFovealStreamBytes = (size_t)192 << 20;   // observations larger than this are streamed (non-temporal stores)

// LDS one workgroup may ask for on this device (160 KiB on gfx950), queried once
static size_t lds_limit() {
    static size_t limit = 0;
    if (limit == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
            limit = (size_t)v;
        else
            limit = 64 * 1024;
    }
    return limit;
}

template <int VARIANT, int MODE, int EPB>
static hipError_t launch_foveal_one(const FovealArgs& a, hipStream_t s) {
    const int cells = a.p.grid * a.p.grid;
    const int L = VARIANT == LMAZE_VARIANT_V1 ? 1 : a.p.n_layouts;
    // obs bit string 32 B + obs_local bit string 16 B + centres 8 B + flags 4 B + reset centre 4 B per env, row masks, layout characters, visit samples
    size_t lds = (size_t)EPB * 64 + (3 * (size_t)L * a.p.grid + 2 * (size_t)a.p.grid) * 8 + (size_t)((L * cells + 15) & ~15);
    if (VARIANT == LMAZE_VARIANT_V4 || VARIANT == LMAZE_VARIANT_V5)
        lds += (size_t)EPB * (2 * W25 * 4 + 8);   // + visit samples, clock, whole-map list
    // envs per workgroup is a performance knob (launch_hint bits 4-7): a size whose LDS does not fit the device falls
    // back to the next smaller one instead of failing the launch (v4-v6 at 256 envs: 164 KiB)
    if constexpr (EPB > 32) {
        if (lds > lds_limit()) return launch_foveal_one<VARIANT, MODE, EPB / 2>(a, s);
    }
    // launch_hint bits 8-9 (plain and fused step): chunks of EPB envs per workgroup - 1 (more than 4, or 16-env chunks: slower)
    const int64_t nchunks = (a.n + EPB - 1) / EPB;
    const int m = MODE == FM_STEP ? ((a.p.launch_hint >> 8) & 3) + 1 : 1;
    const int64_t blocks = (nchunks + m - 1) / m;
    if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
    FovealArgs b = a;
    const int C = VARIANT == LMAZE_VARIANT_V1 ? 4 : (VARIANT == LMAZE_VARIANT_V2 ? 5 : 7);
    b.nt = (size_t)a.n * C * W25 * 4 > kFovealStreamBytes;
    if (LMAZE_XP(a, 2)) b.nt = 0;
    // launch_hint bits 0-3: at most that many workgroups resident per CU, by padding the dynamic LDS (160 KiB per
    // CU), as the step kernel does in its streaming regime (lmaze_step.hip launch_shared); 0 = no cap
    const int per_cu = a.p.launch_hint & 15;
    if (MODE == FM_STEP && per_cu >= 1 && per_cu <= 8) {
        const size_t cap = 160 * 1024;
        const size_t want = ((cap / per_cu + cap / (per_cu + 1)) / 2) & ~(size_t)255;   // between the two thresholds
        if (want > lds && want <= lds_limit()) lds = want;     // per_cu 1 (120 KiB) and 2 (66 KiB) included where the device allows
    }
    const dim3 grid((unsigned)blocks), block(LMAZE_BLOCK);
    if (a.info) {
        char name[96];
        snprintf(name, sizeof(name), "foveal_kernel<v%d, %s, %d, %d, %s>", VARIANT,
                 MODE == FM_STEP ? "step" : (MODE == FM_RESET ? "reset" : (MODE == FM_SETGOAL ? "setgoal" : "planner")), EPB,
                 (a.p.grid == 18 || a.p.grid == 14) ? a.p.grid : 0, a.auto_reset ? "fused-reset" : "plain");
        describe_launch(a.info, name, EPB, (MODE == FM_STEP && per_cu >= 1 && per_cu <= 8 && lds > (size_t)EPB * 64) ? per_cu : 0, m, b.nt != 0,
                        blocks, LMAZE_BLOCK, lds);
        return hipSuccess;
    }
    if constexpr (MODE == FM_STEP) {
        if (a.auto_reset) {
            if (a.p.grid == 18) hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 18, true>), grid, block, lds, s, b);
            else if (a.p.grid == 14) hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 14, true>), grid, block, lds, s, b);
            else hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 0, true>), grid, block, lds, s, b);
            return hipGetLastError();
        }
    }
    if (a.p.grid == 18) hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 18, false>), grid, block, lds, s, b);
    else if (a.p.grid == 14) hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 14, false>), grid, block, lds, s, b);
    else hipLaunchKernelGGL((foveal_kernel<VARIANT, MODE, EPB, 0, false>), grid, block, lds, s, b);
    return hipGetLastError();
}

// launch_hint bits 4-7 (plain step only): envs per workgroup, 2: 32 ... 5: 256; anything else = the default below
template <int VARIANT>
static bool launch_step_hinted(const FovealArgs& a, hipStream_t s, hipError_t& rc) {
    switch ((a.p.launch_hint >> 4) & 15) {
        case 2: rc = launch_foveal_one<VARIANT, FM_STEP, 32>(a, s); return true;
        case 3: rc = launch_foveal_one<VARIANT, FM_STEP, 64>(a, s); return true;
        case 4: rc = launch_foveal_one<VARIANT, FM_STEP, 128>(a, s); return true;
        case 5: rc = launch_foveal_one<VARIANT, FM_STEP, 256>(a, s); return true;
        default: return false;
    }
}

template <int MODE>
static hipError_t launch_foveal_mode(const FovealArgs& a0, hipStream_t s) {
    if (a0.n == 0) return hipSuccess;
    FovealArgs a = a0;
    if (MODE == FM_STEP && a.p.launch_hint == 0 && (!a.auto_reset || a.p.variant == LMAZE_VARIANT_V5 || a.p.variant == LMAZE_VARIANT_V6)) {
        // Default policy of the plain step in the streaming regime (observation larger than the Infinity Cache), as a
        // hint.  Round 2, after workgroups learnt to take several chunks: on a box where every uncapped one-chunk launch of
        // v1 sat at 78.5 us whatever the envs per workgroup (on other boxes 32 envs: 66.8-67.4), 32 envs x 3 chunks ran
        // at 68.0 and x 2 at 68.7; v4 32 x 2 595 against 611; v2 64 envs at 5 workgroups per CU 88.7-89.1 on two
        // boxes against 92-96 uncapped (32 x 2-3: 93-94).
        const int C = a.p.variant == LMAZE_VARIANT_V1 ? 4 : (a.p.variant == LMAZE_VARIANT_V2 ? 5 : 7);
        if ((size_t)a.n * C * W25 * 4 > kFovealStreamBytes) {
            if (a.p.variant == LMAZE_VARIANT_V1) a.p.launch_hint = 0x220;
            else if (a.p.variant == LMAZE_VARIANT_V2) a.p.launch_hint = 0x35;
            // round 3 (window-only visit map): v4 128 envs per workgroup 305.8 us, 64 envs 311.2 (x 2 chunks 321, at 5 per
            // CU 308.7), 32 envs 406; v5/v6 128 envs 446-449 (x 2 chunks 446), 64 envs 503-511
            // (profiles/r03/foveal_sweep_{a,b}.jsonl, two boxes: v4 128 envs x 2 chunks 296.0 / 297.9 us, x 1 304-307, 64 x 2
            // 299.7 / 301.3; v5/v6 128 envs 422.0 / 434.7, x 2 chunks 438.7 / 453.0, 64 envs 503-511)
            else if (a.p.variant == LMAZE_VARIANT_V4) a.p.launch_hint = 0x140;
            else a.p.launch_hint = 0x40;
        }
    }
    if (MODE == FM_STEP && a.p.launch_hint == 0 && a.auto_reset && (size_t)a.n * W25 * 4 * 4 > kFovealStreamBytes) {
        // fused reset (v1, v2, v4; one measured size each, below), same sweeps: v1 at 5 workgroups per CU 74.0 / 74.0 us
        // against 74.2 / 76.8 uncapped; v2 two chunks 97.3 / 99.2 against 102.9 / 100.2; v4 two chunks 402-406 against 426-430
        // (foveal_sweep_ar_a.jsonl, once the envs-per-workgroup hints applied to the fused reset too: v4 128 envs 377-378 us
        // against 417 at 64 envs x 2 chunks; v2 128 x 2 97.3 against 99.3; v1 64 envs at 4-5 per CU 72.9-73.0 against 74.0)
        if (a.p.variant == LMAZE_VARIANT_V1) a.p.launch_hint = 0x35;
        else if (a.p.variant == LMAZE_VARIANT_V2 || a.p.variant == LMAZE_VARIANT_V4) a.p.launch_hint = 0x140;
    }
    if (MODE == FM_STEP) {
        hipError_t rc = hipSuccess;
        switch (a.p.variant) {
            case LMAZE_VARIANT_V1: if (launch_step_hinted<LMAZE_VARIANT_V1>(a, s, rc)) return rc; break;
            case LMAZE_VARIANT_V2: if (launch_step_hinted<LMAZE_VARIANT_V2>(a, s, rc)) return rc; break;
            case LMAZE_VARIANT_V4: if (launch_step_hinted<LMAZE_VARIANT_V4>(a, s, rc)) return rc; break;
            default: if (launch_step_hinted<LMAZE_VARIANT_V5>(a, s, rc)) return rc; break;
        }
    }
    // Defaults, measured at 1M envs with a fresh action row per step (tools/foveal_hint_study.py, tools/foveal_decompose.py;
    // us per step, envs per workgroup 32 / 64 / 128 / 256, round 2, after the render went to bit strings):
    //   v1 (400 B of observation per env)  83 / 78-79 / 84 / 87;  64 envs at 5 workgroups per CU: 73.6-73.7 on three boxes
    //   v2 (500 B)                        119 / 94-96 / 98-104 / 100-106; the cap is flat or worse
    // The bare store loop of these kernels (no set-up, no phase 1) takes 88-93 us on v2 and 72-77 us on v1 whatever
    // the chunk size and the cap: the kernels sit within 3-5 % of what their write pattern -- every workgroup streaming
    // a private, env-aligned chunk -- reaches on this memory system (tools/wbench.hip: 6.1 TB/s for 32-KiB private
    // chunks against 6.9 for a fill in which consecutive workgroups write consecutive 4-KiB pieces).
    // After the set-up went to one barrier with ballot-built row masks (cheap enough for small workgroups), re-measured on
    // two boxes: v1 32 envs per workgroup, uncapped 66.8-67.2 us (0.89 of peak; 64 x 4-5 per CU 71-73, 16 envs 105);
    // v4 32 envs 589-652 against 600-675 with 64 on the same boxes; v2 and v5 stay at 64 (32: 102 / 487 against 88-92 / 441).
    switch (a.p.variant) {
        case LMAZE_VARIANT_V1:
            if (MODE == FM_STEP && !a.auto_reset) return launch_foveal_one<LMAZE_VARIANT_V1, MODE, 32>(a, s);
            return launch_foveal_one<LMAZE_VARIANT_V1, MODE, 64>(a, s);
        case LMAZE_VARIANT_V2:
            if (MODE == FM_STEP && a.auto_reset) return launch_foveal_one<LMAZE_VARIANT_V2, MODE, 256>(a, s);
            return launch_foveal_one<LMAZE_VARIANT_V2, MODE, 64>(a, s);
        case LMAZE_VARIANT_V5:
        case LMAZE_VARIANT_V6:
            return launch_foveal_one<LMAZE_VARIANT_V5, MODE, 64>(a, s);
        default:
            // 32 envs per workgroup: 41 KiB of visit maps streamed + 22 KiB of observation written
            if (MODE == FM_STEP && !a.auto_reset) return launch_foveal_one<LMAZE_VARIANT_V4, MODE, 32>(a, s);
            return launch_foveal_one<LMAZE_VARIANT_V4, MODE, 64>(a, s);
    }
}

static int check_foveal(const LmazeFovealParams* p, const uint8_t* layouts, const LmazeFovealBuffers* b, int64_t n) {
    if (!p || !layouts || !b) return LMAZE_E_NULL;
    const bool v56 = p->variant == LMAZE_VARIANT_V5 || p->variant == LMAZE_VARIANT_V6;
    if (p->variant != LMAZE_VARIANT_V1 && p->variant != LMAZE_VARIANT_V2 && p->variant != LMAZE_VARIANT_V4 && !v56)
        return LMAZE_E_VARIANT;
    if (p->grid < FOV || p->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (p->n_layouts < 1 || p->n_layouts > LMAZE_MAX_LAYOUTS) return LMAZE_E_LAYOUT;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
#ifndef LMAZE_EXPERIMENT
    if (p->launch_hint & ~0x3ff) return LMAZE_E_LAYOUT;
#endif
    if (v56) {
        if (!b->fgoal_xy || !b->foveal_step_count || !b->foveal_reward || !b->foveal_done || !b->visit || !b->ball1_xy ||
            !b->fovea_xy || !b->last_xy || !b->foveal_goal || !b->obs_local)
            return LMAZE_E_NULL;
        if ((uintptr_t)b->obs_local & 15) return LMAZE_E_ALIGN;
    }
    if (!b->ball_xy || !b->step_count || !b->reward || !b->done || !b->obs) return LMAZE_E_NULL;
    if (p->variant == LMAZE_VARIANT_V1 && (!b->fgoal_xy || !b->foveal_step_count || !b->foveal_reward || !b->foveal_done))
        return LMAZE_E_NULL;
    if (p->variant != LMAZE_VARIANT_V1 && (!b->goal_xy || !b->layout_id)) return LMAZE_E_NULL;
    if ((p->variant == LMAZE_VARIANT_V4 || v56) && (!b->visit || !b->visit_clock)) return LMAZE_E_NULL;
    if (((uintptr_t)b->obs & 15) || (b->visit && ((uintptr_t)b->visit & 63))) return LMAZE_E_ALIGN;   // a tile = one 64-byte sector
    return 0;
}

// The reference's float[N,G,G] out of / into the clock-relative tiles (include/lmaze.h): one thread per cell.
__global__ __launch_bounds__(LMAZE_BLOCK) void visit_materialise_kernel(const uint32_t* tiles, const int32_t* clock, float* out,
                                                                        int64_t n, int G) {
    const int64_t i = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x;
    const int CELLS = G * G;
    if (i >= n * CELLS) return;
    const int64_t e = i / CELLS;
    const int c = (int)(i - e * CELLS), x = c / G, y = c - x * G;
    const int TB = visit_tiles(G);
    const uint32_t b = tiles[((size_t)e * TB * TB + (x / VT) * TB + (y / VT)) * (VT * VT) + (x % VT) * VT + (y % VT)];
    out[i] = __uint_as_float(visit_true(b, clock[e] & 0xff));
}

__global__ __launch_bounds__(LMAZE_BLOCK) void visit_load_kernel(uint32_t* tiles, int32_t* clock, const float* in, int64_t n, int G) {
    const int64_t i = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x;
    const int TB = visit_tiles(G), PER = TB * TB * VT * VT;
    if (i >= n * PER) return;
    const int64_t e = i / PER;
    const int r = (int)(i - e * PER), tile = r / (VT * VT), c = r - tile * (VT * VT);
    const int x = (tile / TB) * VT + c / VT, y = (tile % TB) * VT + c % VT;
    tiles[i] = (x < G && y < G) ? __float_as_uint(in[(size_t)e * G * G + x * G + y]) : 0u;
    if (r == 0) clock[e] = VISIT_BIAS;        // stored == true value in this frame; record tag cleared: the next step reads the tiles
}

static FovealArgs make_foveal_args(const LmazeFovealParams* p, const uint8_t* layouts, const LmazeFovealBuffers* b, int64_t n) {
    FovealArgs a;
    a.p = *p;
    a.b = *b;
    a.layouts = layouts;
    a.action = nullptr;
    a.goal2 = nullptr;
    a.mask = nullptr;
    a.n = n;
    a.place = 0;
    a.seed = 0;
    a.epoch = 0;
    a.env_base = 0;
    a.epoch_in = nullptr;
    a.epoch_out = nullptr;
    a.nt = 0;
    a.auto_reset = 0;
    a.info = nullptr;
    return a;
}

}  // namespace lmaze

using namespace lmaze;

extern "C" {

int lmaze_foveal_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                      const LmazeFovealBuffers* bufs, int64_t n, void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (!action) return LMAZE_E_NULL;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.action = action;
    return (int)launch_foveal_mode<FM_STEP>(a, (hipStream_t)stream);
}

int lmaze_foveal_step_autoreset(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                                const LmazeFovealBuffers* bufs, int64_t n, uint64_t seed, uint64_t epoch,
                                int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev, void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V1 && params->variant != LMAZE_VARIANT_V2 && params->variant != LMAZE_VARIANT_V4)
        return LMAZE_E_VARIANT;
    if (!action) return LMAZE_E_NULL;
    if (bad_epoch_words(epoch_in_dev, epoch_out_dev)) return LMAZE_E_ALIGN;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.action = action;
    a.auto_reset = 1;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.epoch_in = epoch_in_dev;
    a.epoch_out = epoch_out_dev;
    return (int)launch_foveal_mode<FM_STEP>(a, (hipStream_t)stream);
}

int lmaze_v5_hier_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                       const int32_t* planner_goal, const LmazeFovealBuffers* bufs, int64_t n, uint64_t seed,
                       uint64_t epoch, int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev,
                       void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V5 && params->variant != LMAZE_VARIANT_V6) return LMAZE_E_VARIANT;
    if (!action || !planner_goal) return LMAZE_E_NULL;
    if (bad_epoch_words(epoch_in_dev, epoch_out_dev)) return LMAZE_E_ALIGN;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.action = action;
    a.goal2 = planner_goal;
    a.auto_reset = 1;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.epoch_in = epoch_in_dev;
    a.epoch_out = epoch_out_dev;
    return (int)launch_foveal_mode<FM_STEP>(a, (hipStream_t)stream);
}

int lmaze_foveal_reset(const LmazeFovealParams* params, const uint8_t* layouts, const uint8_t* mask, int32_t place,
                       uint64_t seed, uint64_t epoch, int64_t env_base, const LmazeFovealBuffers* bufs, int64_t n,
                       void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.mask = mask;
    a.place = place;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    return (int)launch_foveal_mode<FM_RESET>(a, (hipStream_t)stream);
}

int lmaze_v1_set_foveal_goal(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* ij,
                             const uint8_t* mask, const LmazeFovealBuffers* bufs, int64_t n, void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V1) return LMAZE_E_VARIANT;
    if (!ij) return LMAZE_E_NULL;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.action = ij;
    a.mask = mask;
    if (n == 0) return 0;
    return (int)launch_foveal_one<LMAZE_VARIANT_V1, FM_SETGOAL, 64>(a, (hipStream_t)stream);
}

int lmaze_v5_planner_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* goal,
                          const uint8_t* mask, const LmazeFovealBuffers* bufs, int64_t n, void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V5 && params->variant != LMAZE_VARIANT_V6) return LMAZE_E_VARIANT;
    if (!goal) return LMAZE_E_NULL;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.action = goal;
    a.mask = mask;
    return (int)launch_foveal_mode<FM_PLANNER>(a, (hipStream_t)stream);
}

int lmaze_v6_safe_foveal_goal(const LmazeFovealParams* params, const uint8_t* layouts, uint64_t seed, uint64_t epoch,
                              int64_t env_base, const LmazeFovealBuffers* bufs, int32_t* out_goal, int64_t n,
                              void* stream) {
    int rc = check_foveal(params, layouts, bufs, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V5 && params->variant != LMAZE_VARIANT_V6) return LMAZE_E_VARIANT;
    if (!out_goal) return LMAZE_E_NULL;
    if (n == 0) return 0;
    FovealArgs a = make_foveal_args(params, layouts, bufs, n);
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    if (!grid_ok((n + LMAZE_BLOCK - 1) / LMAZE_BLOCK)) return (int)hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(safe_goal_kernel, dim3((unsigned)((n + LMAZE_BLOCK - 1) / LMAZE_BLOCK)), dim3(LMAZE_BLOCK), 0,
                       (hipStream_t)stream, a, out_goal);
    return (int)hipGetLastError();
}

int lmaze_describe_foveal_step(const LmazeFovealParams* params, int64_t n, int32_t auto_reset, char* text_host, int32_t len) {
    if (!params || !text_host || len < 1) return LMAZE_E_NULL;
    const bool v56 = params->variant == LMAZE_VARIANT_V5 || params->variant == LMAZE_VARIANT_V6;
    if (params->variant != LMAZE_VARIANT_V1 && params->variant != LMAZE_VARIANT_V2 && params->variant != LMAZE_VARIANT_V4 && !v56)
        return LMAZE_E_VARIANT;
    if (params->grid < FOV || params->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (params->n_layouts < 1 || params->n_layouts > LMAZE_MAX_LAYOUTS) return LMAZE_E_LAYOUT;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    text_host[0] = 0;
    if (n == 0) return 0;
    LaunchInfo info;
    memset(&info, 0, sizeof(info));
    LmazeFovealBuffers none;
    memset(&none, 0, sizeof(none));
    FovealArgs a = make_foveal_args(params, nullptr, &none, n);     // nothing is dereferenced: the launcher fills `info`
    a.auto_reset = auto_reset ? 1 : 0;
    a.info = &info;
    const int rc = (int)launch_foveal_mode<FM_STEP>(a, nullptr);
    if (rc) return rc;
    snprintf(text_host, (size_t)len, "%s grid=%lld block=%d lds=%lld envs_per_workgroup=%d workgroups_per_cu=%d chunks=%d", info.kernel,
             (long long)info.grid, info.block, (long long)info.lds, info.envs_per_workgroup, info.workgroups_per_cu, info.chunks);
    return 0;
}

int64_t lmaze_foveal_visit_bytes(int32_t grid, int64_t n) {
    if (grid < 1 || grid > LMAZE_MAX_GRID || n < 0) return 0;
    const int64_t tb = visit_tiles(grid);
    return n * (tb * tb * (VT * VT) + VPC) * 4;      // tiles, then one "previous window" record per env
}

static int check_visit(const LmazeFovealParams* p, const LmazeFovealBuffers* b, const void* other, int64_t n) {
    if (!p || !b || !other || !b->visit || !b->visit_clock) return LMAZE_E_NULL;
    if (p->grid < FOV || p->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    if (((uintptr_t)b->visit & 63) || ((uintptr_t)other & 3)) return LMAZE_E_ALIGN;
    return 0;
}

int lmaze_foveal_materialise_visit(const LmazeFovealParams* params, const LmazeFovealBuffers* bufs, float* out, int64_t n,
                                   void* stream) {
    int rc = check_visit(params, bufs, out, n);
    if (rc) return rc;
    if (n == 0) return 0;
    const int64_t blocks = (n * params->grid * params->grid + LMAZE_BLOCK - 1) / LMAZE_BLOCK;
    if (!grid_ok(blocks)) return (int)hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(visit_materialise_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t*>(bufs->visit), bufs->visit_clock, out, n, params->grid);
    return (int)hipGetLastError();
}

int lmaze_foveal_load_visit(const LmazeFovealParams* params, const LmazeFovealBuffers* bufs, const float* in, int64_t n,
                            void* stream) {
    int rc = check_visit(params, bufs, in, n);
    if (rc) return rc;
    if (n == 0) return 0;
    const int64_t tb = visit_tiles(params->grid);
    const int64_t blocks = (n * tb * tb * (VT * VT) + LMAZE_BLOCK - 1) / LMAZE_BLOCK;
    if (!grid_ok(blocks)) return (int)hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(visit_load_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<uint32_t*>(bufs->visit), bufs->visit_clock, in, n, params->grid);
    return (int)hipGetLastError();
}

int lmaze_expand_planes(const float* planes, int32_t channels, int32_t g, int32_t expansion, float* out, int64_t n,
                        void* stream) {
    if (!planes || !out) return LMAZE_E_NULL;
    if (g < 1 || g > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (expansion < 1 || expansion > 16 || channels < 1 || channels > 16) return LMAZE_E_EXPANSION;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    if (((uintptr_t)out & 15) || ((uintptr_t)planes & 3)) return LMAZE_E_ALIGN;
    if (n == 0) return 0;
    ExpandPlanesArgs a;
    a.planes = planes;
    a.out = out;
    a.n = n;
    a.channels = channels;
    a.g = g;
    a.expansion = expansion;
    if (g == FOV && expansion == 7 && !((uintptr_t)out & 63)) {
        // measured (262 144 envs, C = 4/5/7), round 1: 32 KiB x 8 per CU 5.2 TB/s; 48 KiB x 2 per CU + non-temporal 6.2-6.4
        // round 2 (C = 4 / 5 / 7, TB/s): 16 KiB x 3 per CU + non-temporal 6.50 / 6.45 / 6.57, 16 KiB x 4 6.26 / 6.54 / 6.27,
        // 48 KiB x 2 (round 1's choice) 6.48 / 6.37 / 6.23, 32 KiB x 2 6.26 / 6.26 / 6.21
        const int chunk = 4096;
        const int64_t total = n * (int64_t)channels * (FOV * 7) * (FOV * 7);
        const int64_t chunks = (total + chunk - 1) / chunk;
        if (grid_ok(chunks)) {
            size_t lds = ((size_t)(chunk / ((FOV * 7) * (FOV * 7)) + 3) * W25 * 4 + 15) & ~(size_t)15;   // planes touched + one of slack
            if (total * 4 > ((int64_t)192 << 20)) {
                const size_t cap = 160 * 1024, want = ((cap / 3 + cap / 4) / 2) & ~(size_t)255;      // 3 workgroups per CU
                if (want > lds) lds = want;
                hipLaunchKernelGGL((expand_planes_stream_kernel<FOV, 7, true>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds,
                                   (hipStream_t)stream, a, chunk);
            } else {
                hipLaunchKernelGGL((expand_planes_stream_kernel<FOV, 7, false>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds,
                                   (hipStream_t)stream, a, chunk);
            }
            return (int)hipGetLastError();
        }
    }
    const int S = g * expansion;
    const size_t lds = (((size_t)channels * g * g * 4 + (size_t)S * 4) + 15) & ~(size_t)15;
    const unsigned blocks = (unsigned)(n < 65536 ? n : 65536);
    hipLaunchKernelGGL(expand_planes_kernel, dim3(blocks), dim3(LMAZE_BLOCK), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
