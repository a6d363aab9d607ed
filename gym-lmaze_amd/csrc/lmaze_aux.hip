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
        if (!grid_ok((a.n + LMAZE_BLOCK - 1) / LMAZE_BLOCK)) return hipErrorInvalidConfiguration;
        const unsigned blocks = (unsigned)((a.n + LMAZE_BLOCK - 1) / LMAZE_BLOCK);
        const size_t lds = ((size_t)a.grid * a.grid * 2 + 15) & ~(size_t)15;
        if (v3) hipLaunchKernelGGL(reset_shared_kernel<LMAZE_VARIANT_V3>, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
        else hipLaunchKernelGGL(reset_shared_kernel<LMAZE_VARIANT_V0>, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
    } else {
        const int epb = LMAZE_BLOCK / 64;
        if (!grid_ok((a.n + epb - 1) / epb)) return hipErrorInvalidConfiguration;
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
// (lmaze_env.py:217-234, lmaze_env_v3.py:295-301).  Pure write stream: 4*C*(G*E)^2 bytes per env.
// ------------------------------------------------------------------------------------

// Any shape: one workgroup per env, the env's cells and two row/column maps in LDS, lanes stripe the
// env's contiguous C*S*S floats with 16-byte stores, index arithmetic by division.
__global__ __launch_bounds__(LMAZE_BLOCK) void render_expanded_generic_kernel(const ExpandArgs a) {
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
        const int head = (int)(a0 - B), nq = (int)((a1 - a0) >> 2);
        float4* out4 = reinterpret_cast<float4*>(a.out + a0);
        for (int q = tid - 1; q <= nq; q += LMAZE_BLOCK) {           // q = -1 / nq: the ragged head / tail
            const int first = q < 0 ? 0 : head + (q << 2);
            const int count = q < 0 ? head : (q == nq ? (int)(B + L - a1) : 4);
            int c = first / PLANE;
            const int rem = first - c * PLANE;
            int row = rem / S, col = rem - row * S;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < count) v[j] = (cells[rowmap[row] + colmap[col]] & maskl[c]) ? 1.0f : 0.0f;
                if (++col == S) {
                    col = 0;
                    if (++row == S) { row = 0; ++c; }
                }
            }
            if (count == 4) out4[q] = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int j = 0; j < count; ++j) a.out[B + first + j] = v[j];
        }
    }
}

// The reference's shapes (G, E at compile time: the flat index decodes by multiplication).  The output of
// the whole batch is ONE contiguous stream of N*C*S*S floats; workgroup w writes the aligned stretch
// [w*CH, (w+1)*CH) of it, whatever envs that covers (at most two, CH <= one env): an env is 94 864 B at
// 11x11 x7, so per-env workgroups start on 16-byte boundaries only and every 1-KB wave store straddles
// cache lines (5.3 TB/s); aligned 32-KB stretches reach 6 TB/s and more.
// LDS: for each of the two envs, cell bits by (grid row, output column): one read per output float.
template <int GT, int ET, bool NT>
__global__ __launch_bounds__(LMAZE_BLOCK) void render_expanded_stream_kernel(const ExpandArgs a) {
    constexpr int CELLS = GT * GT, S = GT * ET, PLANE = S * S;
    extern __shared__ int4 lds4[];
    int* cells = reinterpret_cast<int*>(lds4);                      // [2][CELLS]: the (at most) two envs of the stretch
    int* maskl = cells + 2 * CELLS;                                 // [LMAZE_MAX_CHANNELS]
    const int C = a.channels, L = C * PLANE, tid = threadIdx.x;
    const int64_t total = a.n * (int64_t)L;
    const int64_t f0 = (int64_t)blockIdx.x * a.chunk_floats;        // first float of this workgroup's stretch
    const int len = (int)min((int64_t)a.chunk_floats, total - f0);
    int64_t env0 = (int64_t)((double)f0 / (double)L);               // floor(f0 / L), fixed up below
    if (env0 * L > f0) --env0;
    if ((env0 + 1) * L <= f0) ++env0;
    const int off0 = (int)(f0 - env0 * L);
    const bool two = off0 + len > L;                                // the stretch runs into env0 + 1
    if (tid < LMAZE_MAX_CHANNELS) maskl[tid] = tid < C ? a.mask[tid] : 0;
    for (int k = tid; k < (two ? 2 * CELLS : CELLS); k += LMAZE_BLOCK) cells[k] = a.obs[(size_t)env0 * CELLS + k];
    __syncthreads();
    float* dst = a.out + f0;
    static_assert(ET >= 4, "four consecutive output columns span at most two cells");
    for (int q = tid; (q << 2) < len; q += LMAZE_BLOCK) {
        int local = off0 + (q << 2);
        const int* cl = cells;
        if (local >= L) { local -= L; cl += CELLS; }
        const int c = local / PLANE;
        const int rem = local - c * PLANE;
        const int row = rem / S, col = rem - row * S;
        // One path for every lane (with S = 77 every wave holds float4s that straddle an output row, so a
        // branch would cost both sides): the cell under the first column, the next cell of the same grid
        // row, and the first cell of the following output row -- which may belong to the next plane / env.
        const int* r0 = cl + (row / ET) * GT;
        const int k0 = col / ET;
        const int m0 = maskl[c];
        const int b0 = r0[k0] & m0, b1 = r0[k0 + 1] & m0;             // r0[GT] is read but never selected
        int row1 = row + 1, c1 = c;
        const int* cl1 = cl;
        if (row1 == S) {
            row1 = 0;
            if (++c1 == C) { c1 = 0; cl1 = cells + CELLS; }              // ragged C*S*S only
        }
        const int bw = cl1[(row1 / ET) * GT] & maskl[c1];
        const int edge = (k0 + 1) * ET;                                // first column of the next cell
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = col + j;
            v[j] = (cj >= S ? bw : (cj >= edge ? b1 : b0)) ? 1.0f : 0.0f;
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
    if (!grid_ok(blocks)) return hipErrorInvalidConfiguration;
    if (src)
        hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, s,
                           static_cast<const int4*>(src), static_cast<int4*>(dst), n16);
    else
        hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)blocks), dim3(LMAZE_BLOCK), 0, s, static_cast<int4*>(dst), n16);
    return hipGetLastError();
}

static hipError_t launch_expand_generic(const ExpandArgs& a, hipStream_t s) {
    const int S = a.grid * a.expansion;
    const unsigned blocks = (unsigned)(a.n < 65536 ? a.n : 65536);
    const size_t lds = (((size_t)a.grid * a.grid * 4 + LMAZE_MAX_CHANNELS * 4 + (size_t)S * 4) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(render_expanded_generic_kernel, dim3(blocks), dim3(LMAZE_BLOCK), lds, s, a);
    return hipGetLastError();
}

// E = 1: the reference's unexpanded planes float32[N,C,G,G] (what it calls retState, lmaze_env.py:208-215) --
// the layout a policy network takes.  Any G.  Same aligned-stretch scheme: the workgroup stages the cells of
// the few envs its stretch covers (contiguous in obs; 16-byte loads when aligned, four dword loads in flight
// per lane otherwise) and every lane turns four consecutive output floats into one 16-byte store.  (Reading
// the cells per lane straight from obs instead: 2.0-3.6 TB/s.)  The two divisions of the index decode are
// multiplications by reciprocals the host computed (exact for the < 2^17 values that occur).
template <bool NT>
__global__ __launch_bounds__(LMAZE_BLOCK) void render_planes_stream_kernel(const ExpandArgs a) {
    extern __shared__ int4 lds4[];
    int* cells = reinterpret_cast<int*>(lds4);                      // [envs of the stretch][CELLS] (+ slack)
    const int G = a.grid, C = a.channels, CELLS = G * G, L = C * CELLS, tid = threadIdx.x;
    const int64_t total = a.n * (int64_t)L;
    const int64_t f0 = (int64_t)blockIdx.x * a.chunk_floats;
    const int len = (int)min((int64_t)a.chunk_floats, total - f0);
    int64_t env0 = (int64_t)((double)f0 / (double)L);               // floor(f0 / L), fixed up below
    if (env0 * L > f0) --env0;
    if ((env0 + 1) * L <= f0) ++env0;
    const int off0 = (int)(f0 - env0 * L);
    const int ne = (int)min((int64_t)((off0 + len + L - 1) / L), a.n - env0);
    __shared__ int maskl[LMAZE_MAX_CHANNELS + 1];
    if (tid <= LMAZE_MAX_CHANNELS) maskl[tid] = tid < C ? a.mask[tid] : 0;
    {
        const int32_t* src = a.obs + (size_t)env0 * CELLS;
        const int nint = ne * CELLS;
        if ((((uintptr_t)src) & 15) == 0) {                          // 16-byte loads when the first env starts aligned
            const int n4 = nint >> 2;
            for (int k = tid; k < n4; k += LMAZE_BLOCK) reinterpret_cast<int4*>(cells)[k] = reinterpret_cast<const int4*>(src)[k];
            for (int k = (n4 << 2) + tid; k < nint; k += LMAZE_BLOCK) cells[k] = src[k];
        } else {
            for (int k = tid; k < nint; k += 4 * LMAZE_BLOCK) {      // four loads in flight per lane
                int v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (k + u * LMAZE_BLOCK < nint) ? src[k + u * LMAZE_BLOCK] : 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u * LMAZE_BLOCK < nint) cells[k + u * LMAZE_BLOCK] = v[u];
            }
        }
    }
    __syncthreads();
    float* dst = a.out + f0;
    for (int q = tid; (q << 2) < len; q += LMAZE_BLOCK) {
        const uint32_t local = (uint32_t)(off0 + (q << 2));
        const uint32_t le = (uint32_t)(((uint64_t)local * a.inv_l) >> 32);           // env of the stretch
        const uint32_t rem = local - le * (uint32_t)L;
        const uint32_t c = (uint32_t)(((uint64_t)rem * a.inv_cells) >> 32);          // plane
        const int cell = (int)(rem - c * (uint32_t)CELLS);
        const bool last = (int)c + 1 == C;
        const int* r0 = cells + le * CELLS;
        const int* r1 = last ? r0 + CELLS : r0;
        const int m0 = maskl[c], m1 = maskl[last ? 0 : c + 1];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = cell + j;
            v[j] = (cj < CELLS ? (r0[cj] & m0) : (r1[cj - CELLS] & m1)) ? 1.0f : 0.0f;
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

// LDS bytes that make exactly `k` workgroups fit a CU's 160 KiB (as for the step kernel, lmaze_step.hip)
static size_t expand_lds_for_workgroups_per_cu(int k) {
    const size_t cap = 160 * 1024;
    return ((cap / k + cap / (k + 1)) / 2) & ~(size_t)255;
}

// Launch policy, measured on MI355X (65 536 envs, 6.2 GB of output; TB/s):
//   stretch x workgroups per CU      32 KiB x 8   32 KiB x 3 + NT   48 KiB x 3 + NT   64 KiB x 3
//   11x11 x7                            5.7            6.2               5.4              6.4 / erratic
//   12x12 x7                            5.4            6.2               5.6              6.3 / erratic
//   18x18 x4                            6.2            6.2               6.0              4.6
//   32x32 x7                            5.3            5.7               6.2              3.8
// The same narrow optimum as the step kernel: ~96 KiB of non-temporal stores in flight per CU.
template <int GT, int ET>
static hipError_t launch_expand_stream(const ExpandArgs& a, hipStream_t s) {
    constexpr int S = GT * ET;
    const int64_t L = (int64_t)a.channels * S * S, total = a.n * L;
    ExpandArgs b = a;
    // Round 2 (65 536 envs, two processes, TB/s; stretch x workgroups per CU, non-temporal): 11x11 x7 16 KiB x 5 6.61,
    // 16 KiB x 6 6.35, 32 KiB x 3 (round 1's choice) 6.2; 12x12 x7 6.56-6.59 / 6.54-6.58 / 6.22; 18x18 x4 6.62 / 6.88-6.92 /
    // 6.2-6.3; 32x32 x7 5.95 / 5.8 / 5.7 against 48 KiB x 3 6.21-6.26.  Plain stores reach 6.6-6.8 on single shapes and
    // fall to 3.6-4.8 on others; 4- and 8-KiB stretches lose to the per-workgroup index set-up (2.0-4.5).
    b.chunk_floats = GT >= 32 ? 12288 : 4096;                        // 48 / 16 KiB per workgroup
    const int per_cu = GT >= 32 ? 3 : (GT == 18 ? 6 : 5);
    if (b.chunk_floats > L) b.chunk_floats = (int32_t)(L & ~(int64_t)1023);   // a stretch covers two envs at most
    if (b.chunk_floats < 1024) return launch_expand_generic(a, s);
    const int64_t chunks = (total + b.chunk_floats - 1) / b.chunk_floats;
    // out must start on a cache line for the stretches to be aligned
    if (!grid_ok(chunks) || ((uintptr_t)a.out & 63)) return launch_expand_generic(a, s);
    size_t lds = ((size_t)2 * GT * GT * 4 + LMAZE_MAX_CHANNELS * 4 + 15) & ~(size_t)15;
    const bool nt = total * 4 > ((int64_t)192 << 20);                // cannot stay in the 256 MiB Infinity Cache
    if (nt) {
        const size_t want = expand_lds_for_workgroups_per_cu(per_cu);
        if (want > lds) lds = want;
        hipLaunchKernelGGL((render_expanded_stream_kernel<GT, ET, true>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds, s, b);
    } else {
        hipLaunchKernelGGL((render_expanded_stream_kernel<GT, ET, false>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds, s, b);
    }
    return hipGetLastError();
}

static hipError_t launch_planes_stream(const ExpandArgs& a, hipStream_t s) {
    const int64_t cells = (int64_t)a.grid * a.grid, L = a.channels * cells, total = a.n * L;
    ExpandArgs b = a;
    b.chunk_floats = 4096;                                           // 16 KiB per workgroup (round 2: 6.14 / 6.40 TB/s at
                                                                     // 1M x 11x11 / 256K x 32x32 against 5.93 / 6.03 with 32 KiB)
    const int64_t chunks = (total + b.chunk_floats - 1) / b.chunk_floats;
    if (cells < 4 || !grid_ok(chunks) || ((uintptr_t)a.out & 63)) return launch_expand_generic(a, s);
    b.inv_l = (((uint64_t)1 << 32) + (uint64_t)L - 1) / (uint64_t)L;
    b.inv_cells = (((uint64_t)1 << 32) + (uint64_t)cells - 1) / (uint64_t)cells;
    size_t lds = ((size_t)(b.chunk_floats / L + 3) * cells * 4 + 15) & ~(size_t)15;
    // read + write stream (the input is 20 % of the traffic): as for the per-env step kernel, capping the
    // occupancy only hurts -- 1M x 11x11: 5.9 TB/s uncapped, 5.0 at 4 workgroups per CU, 4.1 at 3
    if (total * 4 > ((int64_t)192 << 20)) {
        hipLaunchKernelGGL((render_planes_stream_kernel<true>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds, s, b);
    } else {
        hipLaunchKernelGGL((render_planes_stream_kernel<false>), dim3((unsigned)chunks), dim3(LMAZE_BLOCK), lds, s, b);
    }
    return hipGetLastError();
}

hipError_t launch_expand(const ExpandArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (a.expansion == 1) return launch_planes_stream(a, s);
    // the reference's own shapes (v0 12x12 x7, v3 18x18 x4) and BASELINE's grids at x7
    if (a.expansion == 7) {
        if (a.grid == 12) return launch_expand_stream<12, 7>(a, s);
        if (a.grid == 11) return launch_expand_stream<11, 7>(a, s);
        if (a.grid == 8) return launch_expand_stream<8, 7>(a, s);
        if (a.grid == 32) return launch_expand_stream<32, 7>(a, s);
    }
    if (a.expansion == 4 && a.grid == 18) return launch_expand_stream<18, 4>(a, s);
    return launch_expand_generic(a, s);
}

}  // namespace lmaze
