// kbench.hip -- developer tool (not part of the product or the tests): times launch-policy
// variants of the shared-layout step kernel against plain fill kernels in ONE process,
// interleaved rounds (guide rule 24).  Results quoted in DESIGN.md section 5 / lmaze_step.hip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kbench.hip -o tools/kbench
//   ./tools/kbench [rounds=11] [iters=20]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../gym-lmaze_amd/csrc/lmaze_step.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace lmaze;

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(int4* dst, size_t n16, int v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        store16<NT>(dst + i, make_int4(v, v, v, v));
}

// same address schedule as the step kernel: each block owns a contiguous chunk
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunk_kernel(int4* dst, int chunk16, size_t n16, int v) {
    const size_t base = (size_t)blockIdx.x * chunk16;
    for (int q = threadIdx.x; q < chunk16 && base + q < n16; q += 256) store16<NT>(dst + base + q, make_int4(v, v, v, v));
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
};

template <int G, int EPB>
void run_grid(int64_t N, int rounds, int iters, hipStream_t s) {
    const int CELLS = G * G;
    std::vector<uint8_t> lay(CELLS, 'B');
    for (int i = 0; i < G; ++i) lay[i] = lay[(G - 1) * G + i] = lay[i * G] = lay[i * G + G - 1] = 'W';
    lay[1 * G + 1] = 'S';
    lay[(G / 2) * G + G / 2] = 'X';
    std::vector<int32_t> act(N), ball(2 * N);
    srand(1);
    for (int64_t i = 0; i < N; ++i) { act[i] = rand() & 3; ball[2 * i] = 1 + rand() % (G - 2); ball[2 * i + 1] = 1 + rand() % (G - 2); if (ball[2*i]==G/2 && ball[2*i+1]==G/2) ball[2*i]=1; }
    StepArgs a{};
    uint8_t* d_lay; int32_t *d_act, *d_ball, *d_sc, *d_obs; float* d_rew; uint8_t* d_done;
    CK(hipMalloc(&d_lay, 4096)); CK(hipMalloc(&d_act, N * 4)); CK(hipMalloc(&d_ball, N * 8)); CK(hipMalloc(&d_sc, N * 4));
    CK(hipMalloc(&d_rew, N * 4)); CK(hipMalloc(&d_done, N)); CK(hipMalloc(&d_obs, (size_t)N * CELLS * 4));
    CK(hipMemcpy(d_lay, lay.data(), CELLS, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_act, act.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ball, ball.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_sc, 0, N * 4)); CK(hipMemset(d_rew, 0, N * 4)); CK(hipMemset(d_done, 0, N));
    a.layout = d_lay; a.action = d_act; a.ball = (int2*)d_ball; a.step_count = d_sc; a.reward = d_rew;
    a.done = d_done; a.obs = d_obs; a.n = N; a.grid = G; a.step_limit = 100;
    a.reward_wall = -1.f; a.reward_move = -0.01f; a.reward_goal = 100.f;

    const size_t n16 = (size_t)N * CELLS / 4;
    std::vector<Variant> vs;
    auto add = [&](bool nt, int per_cu, int rot, int ar) {
        char nm[128];
        snprintf(nm, sizeof nm, "step G=%d EPB=%d NT=%d wg/CU=%d rot=%d autoreset=%d", G, EPB, (int)nt, per_cu, rot, ar);
        vs.push_back({nm, [=](hipStream_t st) {
            StepArgs b = a; (void)rot; b.auto_reset = ar; b.seed = 1; b.epoch = 5;
            size_t lds = shared_lds_bytes(G, true, EPB);
            if (per_cu < 8) lds = std::max(lds, lds_for_workgroups_per_cu(per_cu));
            const unsigned blocks = (unsigned)((N + EPB - 1) / EPB);
            if (nt) hipLaunchKernelGGL((step_shared_kernel<G, LMAZE_VARIANT_V0, true, EPB, true>), dim3(blocks), dim3(256), lds, st, b);
            else hipLaunchKernelGGL((step_shared_kernel<G, LMAZE_VARIANT_V0, true, EPB, false>), dim3(blocks), dim3(256), lds, st, b);
        }, {}});
    };
    if (N <= (1 << 17)) { add(false, 8, 0, 0); add(true, 8, 0, 0); }   // launch-bound sizes: plain stores, no cap
    else { add(true, 8, 0, 0); add(true, 5, 0, 0); add(true, 4, 0, 0); add(true, 3, 0, 0); add(true, 2, 0, 0); }
    vs.push_back({"fill one-store-per-thread", [=](hipStream_t st) { hipLaunchKernelGGL(fill_kernel<false>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (int4*)d_obs, n16, 3); }, {}});
    vs.push_back({"fill one-store-per-thread, non-temporal", [=](hipStream_t st) { hipLaunchKernelGGL(fill_kernel<true>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (int4*)d_obs, n16, 3); }, {}});
    vs.push_back({"fill chunk-per-workgroup, non-temporal, 3 wg/CU", [=](hipStream_t st) { const int ch = EPB * CELLS / 4; hipLaunchKernelGGL(fill_chunk_kernel<true>, dim3((unsigned)((n16 + ch - 1) / ch)), dim3(256), lds_for_workgroups_per_cu(3), st, (int4*)d_obs, ch, n16, 3); }, {}});
    vs.push_back({"fill chunk-per-workgroup", [=](hipStream_t st) { const int ch = EPB * CELLS / 4; hipLaunchKernelGGL(fill_chunk_kernel<false>, dim3((unsigned)((n16 + ch - 1) / ch)), dim3(256), 0, st, (int4*)d_obs, ch, n16, 3); }, {}});
    vs.push_back({"hipMemsetAsync", [=](hipStream_t st) { (void)hipMemsetAsync(d_obs, 1, n16 * 16, st); }, {}});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) v.launch(s);
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / iters);
        }
    const double step_bytes = (double)N * (37 + 4 * CELLS), fill_bytes = (double)n16 * 16;
    printf("N=%lld G=%d  step bytes %.1f MB, fill bytes %.1f MB\n", (long long)N, G, step_bytes / 1e6, fill_bytes / 1e6);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        const double bytes = v.name.rfind("step", 0) == 0 ? step_bytes : fill_bytes;
        printf("%-52s median %8.2f us  min %8.2f us   %7.1f GB/s\n", v.name.c_str(), med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9);
    }
    hipFree(d_lay); hipFree(d_act); hipFree(d_ball); hipFree(d_sc); hipFree(d_rew); hipFree(d_done); hipFree(d_obs);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 11;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // envs per workgroup x workgroups per CU: the optimum sits at ~192 envs (93 KB of stores) in flight per
    // CU -- 64 x 3 = 79 us, 48 x 4 = 80 us, 96 x 2 = 89 us -- and the chunk must be a multiple of 64 B
    // (56 or 72 envs of 484 B are only 32-B aligned: 108-124 us)
    run_grid<11, 64>(1 << 20, rounds, iters, s);
    if (argc > 3 && argv[3][0] == 'c') {   // "c2": the launch-bound configuration, envs per workgroup
        run_grid<8, 128>(65536, rounds, iters, s);
        run_grid<8, 64>(65536, rounds, iters, s);
        run_grid<8, 32>(65536, rounds, iters, s);
        run_grid<8, 16>(65536, rounds, iters, s);
        return 0;
    }
    if (argc > 3) return 0;   // any other third argument: the metric's shape only
    run_grid<11, 48>(1 << 20, rounds, iters, s);
    run_grid<11, 96>(1 << 20, rounds, iters, s);
    run_grid<8, 128>(1 << 21, rounds, iters, s);
    run_grid<12, 64>(1 << 20, rounds, iters, s);
    run_grid<32, 8>(1 << 17, rounds, iters, s);
    return 0;
}
