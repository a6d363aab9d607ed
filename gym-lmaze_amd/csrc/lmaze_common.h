// Shared device/host definitions for the gfx950 L-maze kernels (not part of the C ABI).
#ifndef LMAZE_COMMON_H_
#define LMAZE_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lmaze.h"

#define LMAZE_BLOCK 256  // threads per workgroup: 4 wave64s, one per SIMD of a CU

namespace lmaze {

// The streaming 16-byte store of every observation writer.  BITS picks the gfx950 cache-policy bits: 0 = the compiler's
// non-temporal store (`nt`), 2 = `sc0 sc1 nt` (system-scope write-through + streaming).  tools/stbench.hip: the bare
// 507-MB fill runs 72.1 us with `sc0 sc1 nt` / `sc1` against 75.6-76.2 with `nt` and 74.2-74.8 plain -- but of the kernels
// only the per-env-layout step follows (1M x 32x32: 834-840 us against 884-888, three interleaved passes); the
// shared-layout step, v1 and v2 lose 3-10 % and v5 20 % with it (profiles/r03/studies/stream_bits.txt), so it is a
// per-kernel choice.  The asm form carries no memory clobber: observations are write-only inside a kernel.
template <int BITS = 0, typename V4>
__device__ __forceinline__ void stream_store16(V4* p, V4 t) {
    static_assert(sizeof(V4) == 16, "one dwordx4");
    if constexpr (BITS == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(t));
    else __builtin_nontemporal_store(t, p);
}


// What a launcher decided, for lmaze_describe_step / lmaze_describe_foveal_step (include/lmaze.h): filled INSTEAD of
// launching when the args carry a pointer to one, by the very code that launches otherwise.
struct LaunchInfo {
    char kernel[96];
    int32_t envs_per_workgroup, workgroups_per_cu, chunks, non_temporal, block;
    int64_t grid, lds;
};
void describe_launch(LaunchInfo* info, const char* kernel, int epb, int per_cu, int chunks, bool nt, int64_t grid, int block, size_t lds);

// Everything a step/observe launch needs, passed by value in the kernarg segment.
struct StepArgs {
    const uint8_t* layout;  // [G*G] shared or [N*G*G] per env, reference cell characters
    const int32_t* action;  // [N]
    int2* ball;             // [N] (x = row, y = column)
    const int2* goal;       // [N] v3 only
    int32_t* step_count;    // [N]
    float* reward;          // [N]
    uint8_t* done;          // [N]
    int32_t* goal_count;    // [N] or null
    int32_t* obs;           // [N*G*G] or null
    uint8_t* obs8;          // narrow observation uint8[N*G*G] (lmaze_step_u8 / lmaze_observe_u8), instead of obs
    int64_t n;
    int32_t grid;           // G when the kernel is not specialised on it
    int32_t step_limit;
    float reward_wall, reward_move, reward_goal;
    int32_t envs_per_block;  // per-env-layout kernels: envs whose layouts one workgroup tiles in LDS
    // fused auto-reset (lmaze_step_*_autoreset): an env whose done flag is set on entry is
    // re-placed exactly as lmaze_reset(mask = done, seed, epoch, env_base) would, then stepped
    int32_t auto_reset;
    uint64_t seed, epoch;
    int64_t env_base;
    // device-resident epoch (captured graphs): the launch draws with epoch + *epoch_in and its workgroup 0
    // leaves *epoch_in + 1 in *epoch_out (a different word: the other workgroups still read epoch_in)
    const uint64_t* epoch_in;
    uint64_t* epoch_out;
    int2* goal_rw;           // v3 + auto_reset: the goal array, writable
    const uint8_t* mask;     // observe only: re-render just the envs with mask != 0 (null = all)
    int32_t launch_hint;     // LmazeParams.launch_hint (0 = library default policy)
    LaunchInfo* info;        // host pointer; non-null: describe the launch instead of queueing it
};

// masked on-device reset (lmaze_aux.hip)
struct ResetArgs {
    const uint8_t* layout;
    const uint8_t* mask;  // null = every env
    int2* ball;
    int2* goal;           // v3 only
    int32_t* step_count;
    float* reward;
    uint8_t* done;
    int64_t n;
    uint64_t seed, epoch;
    int64_t env_base;  // global index of env 0 (sharded batches)
    int32_t grid;
};

// reference-layout xE render (lmaze_aux.hip)
struct ExpandArgs {
    const int32_t* obs;
    float* out;
    int64_t n;
    int32_t grid, expansion, channels;
    int32_t mask[LMAZE_MAX_CHANNELS];
    int32_t chunk_floats;  // stream kernels: floats per workgroup (multiple of 1024; x E: <= one env)
    uint64_t inv_l, inv_cells;  // x1 planes kernel: ceil(2^32 / (C*G*G)), ceil(2^32 / (G*G))
};

// device-resident epoch words of the *_autoreset entry points: 8-byte aligned, epoch_out only together with
// epoch_in, and never the same word (the other workgroups still read epoch_in while workgroup 0 writes)
inline bool bad_epoch_words(const uint64_t* in, const uint64_t* out) {
    return ((((uintptr_t)in) | ((uintptr_t)out)) & 7) != 0 || (in && in == out) || (!in && out);
}

// A launch of `blocks` workgroups of LMAZE_BLOCK threads is accepted by HIP only while blocks * LMAZE_BLOCK
// stays below 2^32; every launcher checks its block count (computed in 64 bits) with this before narrowing it,
// so an over-large env count is refused with hipErrorInvalidConfiguration instead of wrapping the grid.
inline bool grid_ok(int64_t blocks) { return blocks >= 1 && blocks <= (int64_t)0xFFFFFF; }

hipError_t launch_step(int variant, bool do_step, const StepArgs& a, int layout_mode, hipStream_t s);
hipError_t launch_step_u8(int variant, bool do_step, const StepArgs& a, hipStream_t s);
hipError_t launch_rollout(int variant, const StepArgs& a, int layout_mode, const int32_t* actions, int32_t T, float* reward_t,
                          uint8_t* done_t, hipStream_t s);
hipError_t launch_reset(int variant, const ResetArgs& a, int layout_mode, hipStream_t s);
hipError_t launch_expand(const ExpandArgs& a, hipStream_t s);
hipError_t launch_probe(const void* src, void* dst, int64_t bytes, hipStream_t s);
hipError_t launch_episode_stats(const uint8_t* done, const float* reward, const int32_t* step_count,
                                const int32_t* goal_count, float reward_goal, int64_t n, int64_t* out4, hipStream_t s);

// static observation bits of one layout cell (include/lmaze.h LMAZE_OBS_*)
template <int VARIANT>
__device__ __forceinline__ int cell_bits(uint8_t c) {
    if (VARIANT == LMAZE_VARIANT_V3) {
        // lmaze_env_v3.py:166 free = B|S|X; wall bit kept as the complement on 'W'
        return c == 'W' ? LMAZE_OBS_WALL : ((c == 'B' || c == 'S' || c == 'X') ? LMAZE_OBS_FREE : 0);
    }
    // lmaze_env.py:92-107: wall 'W', goal 'X', blank 'B'; 'S' is in no static plane
    return c == 'W' ? LMAZE_OBS_WALL : (c == 'X' ? LMAZE_OBS_GOAL : (c == 'B' ? LMAZE_OBS_FREE : 0));
}

// lmaze_env.py:153-170 (and the ids of lmaze_env_v3.py:236-247's strings)
__device__ __forceinline__ void decode_action(int a, int& ox, int& oy) {
    ox = (a == 1) - (a == 0);
    oy = (a == 3) - (a == 2);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The transition rule itself, on values: c = character of the target cell (bx+ox, by+oy).
// Returns true when v0 counts a goal hit (v0:195).
template <int VARIANT>
__device__ __forceinline__ bool transition_rule(const StepArgs& a, uint8_t c, int ox, int oy, int tx, int ty, int sc,
                                                float r_in, int gx, int gy, int& bx, int& by, float& r, bool& dn) {
    bool hit = false;
    if (VARIANT == LMAZE_VARIANT_V3) {
        r = -0.0f;  // v3:224
        if (c == 'W') {
            r = a.reward_wall;  // v3:252
        } else {
            bx = tx; by = ty;   // v3:258-259
            r = (bx + ox == gx && by + oy == gy) ? a.reward_goal : a.reward_move;  // v3:262-265
        }
        dn = (r == a.reward_goal) || (sc > a.step_limit);  // v3:398
    } else {
        r = r_in;  // sticky: no else branch in v0:172-195
        if (c == 'W') {
            r = a.reward_wall;  // v0:174
        } else if (c == 'B') {
            bx = tx; by = ty;   // v0:180-181
            r = a.reward_move;  // v0:184
        } else if (c == 'X') {
            bx = tx; by = ty;   // v0:190-191
            r = a.reward_goal;  // v0:194
            hit = true;         // v0:195
        }
        dn = (r == a.reward_goal) || (sc == a.step_limit);  // v0:246-249
    }
    return hit;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11)
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// Touch every 64-byte line of [p, p + bytes) once, the work spread over the first `blocks` workgroups of the
// launch (one dword load per line; the sum only keeps the loads alive).  Used at kernel start to pull the
// per-env inputs of a whole step into the memory-side cache in ONE burst of reads: fetched chunk by chunk in
// the middle of the launch, each small read turns the saturated write stream around (lmaze_step.hip).
__device__ __forceinline__ int warm_lines(const void* p, int64_t bytes, int blocks) {
    int acc = 0;
    if (p == nullptr || (int)blockIdx.x >= blocks) return 0;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3;          // dword loads: align down
    const int64_t dwords = (int64_t)((reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes - lo) / 4);
    if (dwords < 1) return 0;
    const int64_t lines = (dwords + 15) / 16;
    const int* q = reinterpret_cast<const int*>(lo);
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += (int64_t)blocks * blockDim.x)
        acc += q[min(l * 16, dwords - 1)];
    return acc;
}

// epoch of this launch: the host's count plus the device-resident one, when the caller keeps one
__device__ __forceinline__ uint64_t launch_epoch(uint64_t epoch, const uint64_t* epoch_in) {
    return epoch_in ? epoch + *epoch_in : epoch;
}

// one thread of the launch hands the next launch its epoch (stream order makes it visible)
__device__ __forceinline__ void pass_epoch_on(const uint64_t* epoch_in, uint64_t* epoch_out) {
    if (epoch_in && epoch_out && blockIdx.x == 0 && threadIdx.x == 0) *epoch_out = *epoch_in + 1;
}

// the reset draw of one env: counter = (global env index, epoch), key = seed
__device__ __forceinline__ uint4 env_draw(uint64_t seed, uint64_t epoch, int64_t env_global) {
    const uint64_t e = (uint64_t)env_global;
    return philox4x32_10(make_uint4((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32)),
                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

// cells the reference's rejection loops accept.  v0:73 ball: not 'W', not 'X'.
// v3:149 goal: not 'W' (the ball list is the same list minus the goal cell, v3:158).
template <int VARIANT>
__device__ __forceinline__ bool spawn_ok(uint8_t c) {
    return VARIANT == LMAZE_VARIANT_V3 ? (c != 'W') : (c != 'W' && c != 'X');
}

__device__ __forceinline__ bool interior(int cell, int G) {
    const int x = cell / G, y = cell - x * G;
    return x >= 1 && x <= G - 2 && y >= 1 && y <= G - 2;
}

// One wave compacts the accepted cells of `lay` (row-major) into list[]; returns the count
// (valid in every lane).  All 64 lanes of the wave must call it.
template <int VARIANT>
__device__ __forceinline__ int wave_build_spawn_list(const uint8_t* lay, int G, int CELLS, uint16_t* list, int lane) {
    int count = 0;
    for (int base = 0; base < CELLS; base += 64) {
        const int cell = base + lane;
        const bool ok = cell < CELLS && interior(cell, G) && spawn_ok<VARIANT>(lay[cell]);
        const unsigned long long m = __ballot(ok);
        if (ok) list[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)cell;
        count += __popcll(m);
    }
    return count;
}

// placement from a compacted list: goal = list[(r.x*count)>>32]; ball = the (r.y*(count-1))>>32-th
// entry skipping the goal (v3) or list[(r.y*count)>>32] (v0).  Cells < 0 mean "leave unchanged".
template <int VARIANT>
__device__ __forceinline__ void place_from_list(const uint16_t* list, int count, uint4 r, int& ball_cell, int& goal_cell) {
    ball_cell = -1;
    goal_cell = -1;
    if (VARIANT == LMAZE_VARIANT_V3) {
        int kg = -1;
        if (count > 0) {
            kg = (int)__umulhi(r.x, (uint32_t)count);
            goal_cell = list[kg];
        }
        if (count > 1) {
            int kb = (int)__umulhi(r.y, (uint32_t)(count - 1));
            kb += (kb >= kg);
            ball_cell = list[kb];
        }
    } else if (count > 0) {
        ball_cell = list[__umulhi(r.y, (uint32_t)count)];
    }
}

__device__ __forceinline__ int kth_set_bit(unsigned long long m, int k) {   // position of the k-th (0-based) set bit
    int pos = 0;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        const unsigned long long low = m & ((1ull << sh) - 1ull);
        const int c = __popcll(low);
        if (k >= c) { k -= c; m >>= sh; pos += sh; } else { m = low; }
    }
    return pos;
}

// placement from the accepted cells as bit masks (bit c of the NS x 64-bit string = cell c accepted): same rule and
// same ranking (row-major) as place_from_list
template <int NS>
__device__ __forceinline__ int kth_cell_of_masks(const unsigned long long (&m)[NS], int k) {
    unsigned long long sel = m[0];
    int base = 0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int c = __popcll(m[j]);
        if (!found) {
            if (k < c) { sel = m[j]; base = j * 64; found = true; }
            else k -= c;
        }
    }
    return found ? base + kth_set_bit(sel, k) : -1;
}

template <int VARIANT, int NS>
__device__ __forceinline__ void place_from_masks(const unsigned long long (&m)[NS], int count, uint4 r, int& ball_cell, int& goal_cell) {
    ball_cell = -1;
    goal_cell = -1;
    if (VARIANT == LMAZE_VARIANT_V3) {
        int kg = -1;
        if (count > 0) {
            kg = (int)__umulhi(r.x, (uint32_t)count);
            goal_cell = kth_cell_of_masks<NS>(m, kg);
        }
        if (count > 1) {
            int kb = (int)__umulhi(r.y, (uint32_t)(count - 1));
            kb += (kb >= kg);
            ball_cell = kth_cell_of_masks<NS>(m, kb);
        }
    } else if (count > 0) {
        ball_cell = kth_cell_of_masks<NS>(m, (int)__umulhi(r.y, (uint32_t)count));
    }
}

// k-th accepted cell (row-major, 0-based) of one layout, found by a whole wave with ballots;
// every lane returns the same cell (-1 if there are fewer than k+1)
template <int VARIANT>
__device__ __forceinline__ int wave_kth_cell(const uint8_t* lay, int G, int CELLS, int k, int lane) {
    int seen = 0, found = -1;
    for (int base = 0; base < CELLS; base += 64) {
        const int cell = base + lane;
        const bool ok = cell < CELLS && interior(cell, G) && spawn_ok<VARIANT>(lay[cell]);
        const unsigned long long m = __ballot(ok);
        const int rank = seen + __popcll(m & ((1ull << lane) - 1ull));
        const unsigned long long hit = __ballot(ok && rank == k);
        if (hit) found = base + __ffsll((long long)hit) - 1;
        seen += __popcll(m);
    }
    return found;
}

template <int VARIANT>
__device__ __forceinline__ int wave_count_cells(const uint8_t* lay, int G, int CELLS, int lane) {
    int count = 0;
    for (int base = 0; base < CELLS; base += 64) {
        const int cell = base + lane;
        const bool ok = cell < CELLS && interior(cell, G) && spawn_ok<VARIANT>(lay[cell]);
        count += __popcll(__ballot(ok));
    }
    return count;
}

// whole-wave placement on one env's own layout (HBM or LDS): same rule as place_from_list
template <int VARIANT>
__device__ __forceinline__ void wave_place(const uint8_t* lay, int G, int CELLS, uint4 r, int lane,
                                           int& ball_cell, int& goal_cell) {
    ball_cell = -1;
    goal_cell = -1;
    const int count = wave_count_cells<VARIANT>(lay, G, CELLS, lane);
    if (VARIANT == LMAZE_VARIANT_V3) {
        int kg = -1;
        if (count > 0) {
            kg = (int)__umulhi(r.x, (uint32_t)count);
            goal_cell = wave_kth_cell<VARIANT>(lay, G, CELLS, kg, lane);
        }
        if (count > 1) {
            int kb = (int)__umulhi(r.y, (uint32_t)(count - 1));
            kb += (kb >= kg);
            ball_cell = wave_kth_cell<VARIANT>(lay, G, CELLS, kb, lane);
        }
    } else if (count > 0) {
        ball_cell = wave_kth_cell<VARIANT>(lay, G, CELLS, (int)__umulhi(r.y, (uint32_t)count), lane);
    }
}

// ---- register-tiled placement: the env's layout sits in registers, lane l holding dword 64*j + l in
// w[j] (cells 4*(64*j + l) .. +3), as in step_perenv_wave_kernel.  Same rule as place_from_list /
// wave_place: accepted cells are ranked in row-major order, which here is (j, lane, byte) order.
template <int VARIANT, int G, int NJ>
__device__ __forceinline__ void lane_spawn_masks(const uint32_t (&w)[NJ], int lane, uint32_t (&ok)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        uint32_t m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cell = ((j * 64 + lane) << 2) + q;
            const int x = cell / G, y = cell - x * G;
            const bool in = x >= 1 && x <= G - 2 && y >= 1 && y <= G - 2;
            m |= (uint32_t)(in && spawn_ok<VARIANT>((uint8_t)(w[j] >> (8 * q)))) << q;
        }
        ok[j] = m;
    }
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// k-th accepted cell (k wave-uniform, 0-based); every lane returns the same cell, -1 if k is too large
template <int NJ>
__device__ __forceinline__ int wave_kth_from_masks(const uint32_t (&ok)[NJ], int k, int lane) {
    int running = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = __popc(ok[j]);
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const int total = __shfl(incl, 63, 64);
        if (k < running + total) {
            const int excl = incl - c;
            const bool mine = k >= running + excl && k < running + incl;
            int r = k - running - excl, cell = -1;
            if (mine) {
                uint32_t m = ok[j];
                for (; r > 0; --r) m &= m - 1;  // drop the r lowest set bits
                cell = ((j * 64 + lane) << 2) + (__ffs((int)m) - 1);
            }
            const unsigned long long who = __ballot(mine);
            return __shfl(cell, __ffsll((long long)who) - 1, 64);
        }
        running += total;
    }
    return -1;
}

template <int VARIANT, int G, int NJ>
__device__ __forceinline__ void wave_place_regs(const uint32_t (&w)[NJ], uint4 r, int lane, int& ball_cell, int& goal_cell) {
    uint32_t ok[NJ];
    lane_spawn_masks<VARIANT, G, NJ>(w, lane, ok);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) cnt += __popc(ok[j]);
    const int count = wave_sum(cnt);
    ball_cell = -1;
    goal_cell = -1;
    if (VARIANT == LMAZE_VARIANT_V3) {
        int kg = -1;
        if (count > 0) {
            kg = (int)__umulhi(r.x, (uint32_t)count);
            goal_cell = wave_kth_from_masks<NJ>(ok, kg, lane);
        }
        if (count > 1) {
            int kb = (int)__umulhi(r.y, (uint32_t)(count - 1));
            kb += (kb >= kg);
            ball_cell = wave_kth_from_masks<NJ>(ok, kb, lane);
        }
    } else if (count > 0) {
        ball_cell = wave_kth_from_masks<NJ>(ok, (int)__umulhi(r.y, (uint32_t)count), lane);
    }
}

// OR `bit` into component d (0..3) of v; other d leave v unchanged
__device__ __forceinline__ void or_at(int4& v, int d, int bit) {
    v.x |= (d == 0) ? bit : 0;
    v.y |= (d == 1) ? bit : 0;
    v.z |= (d == 2) ? bit : 0;
    v.w |= (d == 3) ? bit : 0;
}

}  // namespace lmaze
#endif
