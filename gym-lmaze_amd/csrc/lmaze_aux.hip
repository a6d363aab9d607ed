// lmaze_aux.hip -- the kernels either side of the step path: masked on-device reset
// (reference reset(): lmaze_env.py:64-110, lmaze_env_v3.py:134-167) and the
// reference-layout x E nearest-neighbour render (lmaze_env.py:217-234).
#include "lmaze_common.h"

namespace lmaze {


__device__ __forceinline__ void write_reset(const ResetArgs& a, int64_t e) {
    a.step_count[e] = 0;  // v0:110
    a.reward[e] = -0.0f;  // v0:109
    a.done[e] = 0;
}

// Shared layout: wave 0 compacts the accepted cells (row-major) into LDS once per
// workgroup, then one lane per env indexes the list.
template <int VARIANT>
__global__ __launch_bounds__(LMAZE_BLOCK) void reset_shared_kernel(const ResetArgs a) {
    extern __shared__ int4 lds4[];
    uint16_t* list = reinterpret_cast<uint16_t*>(lds4);
    __shared__ int count_s;
    const int G = a.grid, CELLS = G * G;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int count = wave_build_spawn_list<VARIANT>(a.layout, G, CELLS, list, tid);
        if (tid == 0) count_s = count;
    }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * LMAZE_BLOCK + tid;
    if (e >= a.n) return;
    if (a.mask && !a.mask[e]) return;
    int ball_cell, goal_cell;
    place_from_list<VARIANT>(list, count_s, env_draw(a.seed, a.epoch, a.env_base + e), ball_cell, goal_cell);
    if (goal_cell >= 0) a.goal[e] = make_int2(goal_cell / G, goal_cell % G);
    if (ball_cell >= 0) a.ball[e] = make_int2(ball_cell / G, ball_cell % G);
    write_reset(a, e);
}

// Per-env layouts: one wave per env scans that env's G*G bytes (coalesced) with ballots.
template <int VARIANT>
__global__ __launch_bounds__(LMAZE_BLOCK) void reset_perenv_kernel(const ResetArgs a) {
    const int G = a.grid, CELLS = G * G;
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (LMAZE_BLOCK / 64) + (threadIdx.x >> 6);
    if (e >= a.n) return;
    if (a.mask && !a.mask[e]) return;
    int ball_cell, goal_cell;
    wave_place<VARIANT>(a.layout + (size_t)e * CELLS, G, CELLS, env_draw(a.seed, a.epoch, a.env_base + e), lane,
                        ball_cell, goal_cell);
    if (lane == 0) {
        if (goal_cell >= 0) a.goal[e] = make_int2(goal_cell / G, goal_cell % G);
        if (ball_cell >= 0) a.ball[e] = make_int2(ball_cell / G, ball_cell % G);
        write_reset(a, e);
    }
}

// Per-env layouts with G*G a multiple of 256: the layout is read once as dwords into registers
// (256 B per wave instruction) and ranked there (wave_place_regs), one wave per env.
template <int VARIANT, int G>
__global__ __launch_bounds__(LMAZE_BLOCK) void reset_perenv_wave_kernel(const ResetArgs a) {
    constexpr int CELLS = G * G, NJ = CELLS / 256;
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (LMAZE_BLOCK / 64) + (threadIdx.x >> 6);
    if (e >= a.n) return;
    if (a.mask && !a.mask[e]) return;
    const uint32_t* lay32 = reinterpret_cast<const uint32_t*>(a.layout + (size_t)e * CELLS);
    uint32_t w[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = lay32[j * 64 + lane];
    int ball_cell, goal_cell;
    wave_place_regs<VARIANT, G, NJ>(w, env_draw(a.seed, a.epoch, a.env_base + e), lane, ball_cell, goal_cell);
    if (lane == 0) {
        if (goal_cell >= 0) a.goal[e] = make_int2(goal_cell / G, goal_cell % G);
        if (ball_cell >= 0) a.ball[e] = make_int2(ball_cell / G, ball_cell % G);
        write_reset(a, e);
    }
}

template <int G>
static bool launch_reset_wave(bool v3, const ResetArgs& a, unsigned blocks, hipStream_t s) {
    if (v3) hipLaunchKernelGGL((reset_perenv_wave_kernel<LMAZE_VARIANT_V3, G>), dim3(blocks), dim3(LMAZE_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((reset_perenv_wave_kernel<LMAZE_VARIANT_V0, G>), dim3(blocks), dim3(LMAZE_BLOCK), 0, s, a);
    return true;
}

hipError_t launch_reset(int variant, const ResetArgs& a, int layout_mode, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const bool v3 = variant == LMAZE_VARIANT_V3;
    if (layout_mode == LMAZE_LAYOUT_SHARED) {
        const unsigned blocks = (unsigned)((a.n + LMAZE_BLOCK - 1) / LMAZE_BLOCK);
        const size_t lds = ((size_t)a.grid * a.grid * 2 + 15) & ~(size_t)15;
        if (v3) hipLaunchKernelGGL(reset_shared_kernel<LMAZE_VARIANT_V3>, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
        else hipLaunchKernelGGL(reset_shared_kernel<LMAZE_VARIANT_V0>, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
    } else {
        const int epb = LMAZE_BLOCK / 64;
        const unsigned blocks = (unsigned)((a.n + epb - 1) / epb);
        const bool wave = (a.grid == 16 && launch_reset_wave<16>(v3, a, blocks, s)) ||
                          (a.grid == 32 && launch_reset_wave<32>(v3, a, blocks, s)) ||
                          (a.grid == 48 && launch_reset_wave<48>(v3, a, blocks, s)) ||
                          (a.grid == 64 && launch_reset_wave<64>(v3, a, blocks, s));
        if (wave) return hipGetLastError();
        if (v3) hipLaunchKernelGGL(reset_perenv_kernel<LMAZE_VARIANT_V3>, dim3(blocks), dim3(LMAZE_BLOCK), 0, s, a);
        else hipLaunchKernelGGL(reset_perenv_kernel<LMAZE_VARIANT_V0>, dim3(blocks), dim3(LMAZE_BLOCK), 0, s, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Reference-layout render: out[i, c, x*E+xx, y*E+yy] = float((obs[i,x,y] & mask[c]) != 0)
// One workgroup per env; the env's G*G ints and two row/column maps sit in LDS; lanes
// stripe the env's contiguous C*S*S floats with 16-byte stores.
// ------------------------------------------------------------------------------------

__global__ __launch_bounds__(LMAZE_BLOCK) void render_expanded_kernel(const ExpandArgs a) {
    extern __shared__ int4 lds4[];
    const int G = a.grid, E = a.expansion, C = a.channels;
    const int CELLS = G * G, S = G * E, PLANE = S * S, L = C * PLANE;
    int* cells = reinterpret_cast<int*>(lds4);                      // [CELLS]
    int* maskl = cells + CELLS;                                     // [LMAZE_MAX_CHANNELS]
    uint16_t* rowmap = reinterpret_cast<uint16_t*>(maskl + LMAZE_MAX_CHANNELS);  // [S] row -> (row / E) * G
    uint16_t* colmap = rowmap + S;                                  // [S] col -> col / E
    const int tid = threadIdx.x;
    if (tid < LMAZE_MAX_CHANNELS) maskl[tid] = tid < C ? a.mask[tid] : 0;
    for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        __syncthreads();
        for (int k = tid; k < CELLS; k += LMAZE_BLOCK) cells[k] = a.obs[(size_t)i * CELLS + k];
        for (int k = tid; k < S; k += LMAZE_BLOCK) {
            rowmap[k] = (uint16_t)((k / E) * G);
            colmap[k] = (uint16_t)(k / E);
        }
        __syncthreads();
        const size_t B = (size_t)i * L;
        const size_t a0 = (B + 3) & ~(size_t)3, a1 = (B + L) & ~(size_t)3;
        auto value = [&](int local) -> float {
            const int c = local / PLANE;
            const int rem = local - c * PLANE;
            const int row = rem / S, col = rem - row * S;
            return (cells[rowmap[row] + colmap[col]] & maskl[c]) ? 1.0f : 0.0f;
        };
        // ragged head/tail (only when C*S*S is not a multiple of 4)
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
                v[j] = (cells[rowmap[row] + colmap[col]] & maskl[c]) ? 1.0f : 0.0f;
                if (++col == S) {
                    col = 0;
                    if (++row == S) { row = 0; ++c; }
                }
            }
            out4[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ------------------------------------------------------------------------------------
// Episode statistics: four integer sums over the batch (grid-stride, wave shuffles, one 64-bit
// atomic per workgroup and counter)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LMAZE_BLOCK) void episode_stats_kernel(const uint8_t* done, const float* reward,
                                                                    const int32_t* step_count, const int32_t* goal_count,
                                                                    float reward_goal, int64_t n,
                                                                    unsigned long long* out4) {
    long long acc[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * LMAZE_BLOCK) {
        const bool d = done[i] != 0;
        acc[0] += d;
        acc[1] += reward[i] == reward_goal;
        acc[2] += d ? step_count[i] : 0;
        if (goal_count) acc[3] += goal_count[i];
    }
    __shared__ long long part[4][LMAZE_BLOCK / 64];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long long v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        long long v = 0;
        for (int w = 0; w < LMAZE_BLOCK / 64; ++w) v += part[threadIdx.x][w];
        if (v) atomicAdd(out4 + threadIdx.x, (unsigned long long)v);
    }
}

hipError_t launch_episode_stats(const uint8_t* done, const float* reward, const int32_t* step_count,
                                const int32_t* goal_count, float reward_goal, int64_t n, int64_t* out4, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out4, 0, 4 * sizeof(int64_t), s);
    if (e != hipSuccess || n == 0) return e;
    int64_t blocks = (n + LMAZE_BLOCK - 1) / LMAZE_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(episode_stats_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, s, done, reward, step_count,
                       goal_count, reward_goal, n, reinterpret_cast<unsigned long long*>(out4));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// measured ceilings of the box ("a measured fill/copy-kernel ceiling on the same box"):
// ONE 16-byte access per thread in launch order, the pattern that reaches the highest write rate here.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LMAZE_BLOCK) void probe_fill_kernel(int4* dst, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x;
    if (i < n16) dst[i] = make_int4((int)i, 1, 2, 3);
}

__global__ __launch_bounds__(LMAZE_BLOCK) void probe_copy_kernel(const int4* __restrict__ src, int4* __restrict__ dst,
                                                                 int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * LMAZE_BLOCK + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

hipError_t launch_probe(const void* src, void* dst, int64_t bytes, hipStream_t s) {
    const int64_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const int64_t blocks = (n16 + LMAZE_BLOCK - 1) / LMAZE_BLOCK;
    if (src)
        hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, s,
                           static_cast<const int4*>(src), static_cast<int4*>(dst), n16);
    else
        hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, s, static_cast<int4*>(dst), n16);
    return hipGetLastError();
}

hipError_t launch_expand(const ExpandArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const int S = a.grid * a.expansion;
    const size_t lds = (((size_t)a.grid * a.grid * 4 + LMAZE_MAX_CHANNELS * 4 + (size_t)S * 4) + 15) & ~(size_t)15;
    const unsigned blocks = (unsigned)(a.n < 65536 ? a.n : 65536);
    hipLaunchKernelGGL(render_expanded_kernel, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
    return hipGetLastError();
}

}  // namespace lmaze
