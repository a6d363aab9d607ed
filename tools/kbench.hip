// kbench.hip -- developer tool (not part of the product or the tests): times variants of the
// step kernel against plain fill kernels in ONE process, interleaved rounds (guide rule 24).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kbench.hip -o tools/kbench
//   ./tools/kbench [N=1048576] [rounds=15] [iters=20]
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

template <int TPB, bool NT>
__global__ __launch_bounds__(TPB) void fill_chunk_tpb_kernel(int4* dst, int chunk16, size_t n16, int v) {
    const size_t base = (size_t)blockIdx.x * chunk16;
    for (int q = threadIdx.x; q < chunk16 && base + q < n16; q += TPB) store16<NT>(dst + base + q, make_int4(v, v, v, v));
}

// interleaved super-chunks: BP consecutive blocks stream one region of BP*I pieces together;
// block b' writes pieces b' + i*BP (piece = PIECE16 x 16 B, one store per thread per iteration)
template <bool NT>
__global__ __launch_bounds__(256) void fill_interleaved_kernel(int4* dst, int BP, int I, int PIECE16, size_t n16, int v) {
    const size_t sc = blockIdx.x / BP;
    const int bp = blockIdx.x % BP;
    if ((int)threadIdx.x >= PIECE16) return;
    for (int i = 0; i < I; ++i) {
        const size_t piece = (sc * I + i) * BP + bp;
        const size_t idx = piece * PIECE16 + threadIdx.x;
        if (idx < n16) store16<NT>(dst + idx, make_int4(v, v, v, v));
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    constexpr int G = 11;
    const int CELLS = G * G;
    hipStream_t s;
    CK(hipStreamCreate(&s));

    std::vector<uint8_t> lay(CELLS, 'B');
    for (int i = 0; i < G; ++i) lay[i] = lay[(G - 1) * G + i] = lay[i * G] = lay[i * G + G - 1] = 'W';
    lay[1 * G + 1] = 'S';
    lay[5 * G + 5] = 'X';
    std::vector<int32_t> act(N), ball(2 * N);
    srand(1);
    for (int64_t i = 0; i < N; ++i) { act[i] = rand() & 3; ball[2 * i] = 1 + rand() % (G - 2); ball[2 * i + 1] = 1 + rand() % (G - 2); if (ball[2*i]==5 && ball[2*i+1]==5) ball[2*i]=4; }

    StepArgs a{};
    uint8_t* d_lay; int32_t *d_act, *d_ball, *d_sc, *d_obs; float* d_rew; uint8_t* d_done;
    CK(hipMalloc(&d_lay, 256)); CK(hipMalloc(&d_act, N * 4)); CK(hipMalloc(&d_ball, N * 8)); CK(hipMalloc(&d_sc, N * 4));
    CK(hipMalloc(&d_rew, N * 4)); CK(hipMalloc(&d_done, N)); CK(hipMalloc(&d_obs, (size_t)N * CELLS * 4));
    CK(hipMemcpy(d_lay, lay.data(), CELLS, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_act, act.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ball, ball.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_sc, 0, N * 4)); CK(hipMemset(d_rew, 0, N * 4));
    a.layout = d_lay; a.action = d_act; a.ball = (int2*)d_ball; a.goal = nullptr; a.step_count = d_sc; a.reward = d_rew;
    a.done = d_done; a.goal_count = nullptr; a.obs = d_obs; a.n = N; a.grid = G; a.step_limit = 100;
    a.reward_wall = -1.f; a.reward_move = -0.01f; a.reward_goal = 100.f;

    const size_t n16 = (size_t)N * CELLS / 4;
    std::vector<Variant> vs;
#define STEP_VARIANT(EPB, NT) vs.push_back({"step EPB=" #EPB " NT=" #NT, [=](hipStream_t st) { \
        const unsigned blocks = (unsigned)((N + EPB - 1) / EPB); \
        hipLaunchKernelGGL((step_shared_kernel<G, LMAZE_VARIANT_V0, true, EPB, NT>), dim3(blocks), dim3(256), \
                           shared_lds_bytes(G, true, EPB), st, a); }, {}})
    STEP_VARIANT(256, false);
    STEP_VARIANT(256, true);
    STEP_VARIANT(128, false);
    STEP_VARIANT(128, true);
    STEP_VARIANT(64, false);
    STEP_VARIANT(64, true);
    STEP_VARIANT(32, false);
    STEP_VARIANT(512, false);
    STEP_VARIANT(1024, false);
    vs.push_back({"fill grid=2048 plain", [=](hipStream_t st) { hipLaunchKernelGGL(fill_kernel<false>, dim3(2048), dim3(256), 0, st, (int4*)d_obs, n16, 3); }, {}});
    vs.push_back({"fill grid=2048 NT", [=](hipStream_t st) { hipLaunchKernelGGL(fill_kernel<true>, dim3(2048), dim3(256), 0, st, (int4*)d_obs, n16, 3); }, {}});
    vs.push_back({"fill grid=full plain", [=](hipStream_t st) { hipLaunchKernelGGL(fill_kernel<false>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (int4*)d_obs, n16, 3); }, {}});
    vs.push_back({"fill chunk=7744 plain", [=](hipStream_t st) { hipLaunchKernelGGL(fill_chunk_kernel<false>, dim3((unsigned)((n16 + 7743) / 7744)), dim3(256), 0, st, (int4*)d_obs, 7744, n16, 3); }, {}});
    vs.push_back({"fill chunk=7744 NT", [=](hipStream_t st) { hipLaunchKernelGGL(fill_chunk_kernel<true>, dim3((unsigned)((n16 + 7743) / 7744)), dim3(256), 0, st, (int4*)d_obs, 7744, n16, 3); }, {}});
    vs.push_back({"fill chunk=1936 plain", [=](hipStream_t st) { hipLaunchKernelGGL(fill_chunk_kernel<false>, dim3((unsigned)((n16 + 1935) / 1936)), dim3(256), 0, st, (int4*)d_obs, 1936, n16, 3); }, {}});
#define FILLC(TPB, CH, NT) vs.push_back({"fillc TPB=" #TPB " chunk16=" #CH " NT=" #NT, [=](hipStream_t st) { \
        hipLaunchKernelGGL((fill_chunk_tpb_kernel<TPB, NT>), dim3((unsigned)((n16 + CH - 1) / CH)), dim3(TPB), 0, st, (int4*)d_obs, CH, n16, 3); }, {}})
    FILLC(256, 256, false); FILLC(256, 512, false); FILLC(256, 1024, false); 
    
    
    
#define FILLI(BP, I, PC, NT) vs.push_back({"filli BP=" #BP " I=" #I " piece16=" #PC " NT=" #NT, [=](hipStream_t st) { \
        const size_t pieces = (n16 + PC - 1) / PC; const unsigned blocks = (unsigned)((pieces + I - 1) / I); \
        hipLaunchKernelGGL((fill_interleaved_kernel<NT>), dim3(blocks), dim3(256), 0, st, (int4*)d_obs, BP, I, PC, n16, 3); }, {}})
    FILLI(8, 30, 256, false); FILLI(16, 30, 256, false); FILLI(64, 30, 256, false); FILLI(256, 30, 256, false); FILLI(1024, 30, 256, false);
    FILLI(64, 8, 256, false); FILLI(256, 8, 256, false); FILLI(2048, 8, 256, false);
    FILLI(64, 30, 242, false); FILLI(256, 30, 242, false); FILLI(64, 30, 256, true);
    vs.push_back({"hipMemsetAsync", [=](hipStream_t st) { (void)hipMemsetAsync(d_obs, 1, n16 * 16, st); }, {}});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) { v.launch(s); }
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
        printf("%-28s median %8.2f us  min %8.2f us   %7.1f GB/s (median)\n", v.name.c_str(), med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9);
    }
    return 0;
}
