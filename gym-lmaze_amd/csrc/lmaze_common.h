// Shared device/host definitions for the gfx950 L-maze kernels (not part of the C ABI).
#ifndef LMAZE_COMMON_H_
#define LMAZE_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lmaze.h"

#define LMAZE_BLOCK 256  // threads per workgroup: 4 wave64s, one per SIMD of a CU

namespace lmaze {

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
    int64_t n;
    int32_t grid;           // G when the kernel is not specialised on it
    int32_t step_limit;
    float reward_wall, reward_move, reward_goal;
    int32_t envs_per_block;  // per-env-layout kernels: envs whose layouts one workgroup tiles in LDS
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
};

hipError_t launch_step(int variant, bool do_step, const StepArgs& a, int layout_mode, hipStream_t s);
hipError_t launch_reset(int variant, const ResetArgs& a, int layout_mode, hipStream_t s);
hipError_t launch_expand(const ExpandArgs& a, hipStream_t s);

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

// One env's transition.  `lay` is that env's layout (LDS).  Returns the new ball cell index.
// Out-of-range coordinates cannot occur with a 'W'-bordered layout (the reference would
// raise IndexError or wrap); indices are clamped only so a bad input cannot fault the GPU.
template <int VARIANT>
__device__ __forceinline__ void transition(const StepArgs& a, const uint8_t* lay, int G, int64_t e,
                                           int& bx, int& by, int gx, int gy) {
    const int act = a.action[e];
    const int sc = a.step_count[e] + 1;  // v0:151, v3:225
    int ox, oy;
    decode_action(act, ox, oy);
    const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
    const uint8_t c = lay[tx * G + ty];  // v0:172, v3:251
    float r;
    bool dn;
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
        r = a.reward[e];  // sticky: no else branch in v0:172-195
        if (c == 'W') {
            r = a.reward_wall;  // v0:174
        } else if (c == 'B') {
            bx = tx; by = ty;   // v0:180-181
            r = a.reward_move;  // v0:184
        } else if (c == 'X') {
            bx = tx; by = ty;   // v0:190-191
            r = a.reward_goal;  // v0:194
            if (a.goal_count) a.goal_count[e] += 1;  // v0:195
        }
        dn = (r == a.reward_goal) || (sc == a.step_limit);  // v0:246-249
    }
    a.ball[e] = make_int2(bx, by);
    a.step_count[e] = sc;
    a.reward[e] = r;
    a.done[e] = dn ? 1 : 0;
}

// The same transition read off the LDS pattern of static plane bits instead of the layout
// characters: WALL <=> 'W'; v0 FREE <=> 'B', GOAL <=> 'X', none <=> 'S' (no branch fires).
template <int VARIANT>
__device__ __forceinline__ void transition_bits(const StepArgs& a, const int* pat, int G, int64_t e,
                                                int& bx, int& by, int gx, int gy) {
    const int act = a.action[e];
    const int sc = a.step_count[e] + 1;  // v0:151, v3:225
    int ox, oy;
    decode_action(act, ox, oy);
    const int tx = clampi(bx + ox, 0, G - 1), ty = clampi(by + oy, 0, G - 1);
    const int cb = pat[tx * G + ty];  // v0:172, v3:251
    float r;
    bool dn;
    if (VARIANT == LMAZE_VARIANT_V3) {
        r = -0.0f;  // v3:224
        if (cb & LMAZE_OBS_WALL) {
            r = a.reward_wall;  // v3:252
        } else {
            bx = tx; by = ty;   // v3:258-259
            r = (bx + ox == gx && by + oy == gy) ? a.reward_goal : a.reward_move;  // v3:262-265
        }
        dn = (r == a.reward_goal) || (sc > a.step_limit);  // v3:398
    } else {
        r = a.reward[e];  // sticky: no else branch in v0:172-195
        if (cb & LMAZE_OBS_WALL) {
            r = a.reward_wall;  // v0:174
        } else if (cb & LMAZE_OBS_FREE) {
            bx = tx; by = ty;   // v0:180-181
            r = a.reward_move;  // v0:184
        } else if (cb & LMAZE_OBS_GOAL) {
            bx = tx; by = ty;   // v0:190-191
            r = a.reward_goal;  // v0:194
            if (a.goal_count) a.goal_count[e] += 1;  // v0:195
        }
        dn = (r == a.reward_goal) || (sc == a.step_limit);  // v0:246-249
    }
    a.ball[e] = make_int2(bx, by);
    a.step_count[e] = sc;
    a.reward[e] = r;
    a.done[e] = dn ? 1 : 0;
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
