// wbench.hip -- developer tool (not part of the product or the tests): which WRITE patterns reach the fill rate
// on MI355X.  The foveal step kernels are write streams (400-700 B of observation per env against <= 30 B of
// state); their bare store loop plateaus at 5.5 TB/s where a one-store-per-thread fill reaches 6.8.  This times
// the patterns in between, ONE process, interleaved rounds, on a buffer of the v2 observation's size.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/wbench.hip -o tools/wbench && ./tools/wbench [rounds] [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(float* p, size_t q, v4f v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p) + q);
    else reinterpret_cast<v4f*>(p)[q] = v;
}

// MODE 0: the workgroup stripes its chunk, 4 KiB per iteration (the step kernels' schedule)
// MODE 1: each wave owns a contiguous quarter of the chunk, 1 KiB per iteration
// MODE 5: wave-autonomous pieces (see the kernel)
// MODE 3: MODE 0, but workgroup b starts at iteration (b * grp) mod iterations of its chunk and wraps
// MODE 2: chunk interleaved over a group of `grp` workgroups: workgroup w of the group writes the 4-KiB pieces
//         w, w + grp, w + 2 grp ... of the group's grp*chunk bytes (the group's front stays compact)
// pace: s_sleep between two stores of a lane
template <bool NT, int MODE>
__global__ __launch_bounds__(256) void chunk_kernel(float* dst, int chunk16, size_t n16, int grp, int pace, const int* dep) {
    extern __shared__ int4 pad[];
    const int tid = threadIdx.x;
    float seed = 1.0f;
    if (dep) seed = (float)dep[(size_t)(blockIdx.x & 0x3ffff) * 64 + (tid & 63)];      // a dependent load in front of the stores
    const v4f v = {seed, 0.f, 1.f, 0.f};
    if (MODE == 0) {
        const size_t base = (size_t)blockIdx.x * chunk16;
        for (int q = tid; q < chunk16 && base + q < n16; q += 256) {
            st16<NT>(dst, base + q, v);
            if (pace) __builtin_amdgcn_s_sleep(1);
        }
    } else if (MODE == 5) {
        // wave-autonomous pieces: global wave w writes pieces w, w + W, w + 2W ... (W = waves of the grid) of `chunk16`
        // 16-byte stores each (a piece = 8 envs of 11x11 = 242 stores, ...), `grp` pieces per wave, a dependent load first
        const int wpb = blockDim.x >> 6, lane = tid & 63;
        const size_t W = (size_t)gridDim.x * wpb, w0 = (size_t)blockIdx.x * wpb + (tid >> 6);
        for (int k = 0; k < grp; ++k) {
            const size_t base = (w0 + (size_t)k * W) * chunk16;
            for (int q = lane; q < chunk16 && base + q < n16; q += 64) st16<NT>(dst, base + q, v);
        }
    } else if (MODE == 3) {                                                  // MODE 0 with the start rotated per workgroup (grp = multiplier)
        const size_t base = (size_t)blockIdx.x * chunk16;
        const int iters = (chunk16 + 255) >> 8;
        const int rot = (int)((blockIdx.x * (unsigned)grp) % (unsigned)iters);
        for (int it = 0; it < iters; ++it) {
            int k = it + rot;
            if (k >= iters) k -= iters;
            const int q = k * 256 + tid;
            if (q < chunk16 && base + q < n16) st16<NT>(dst, base + q, v);
        }
    } else if (MODE == 1) {
        const int per_wave = chunk16 >> 2;
        const size_t base = (size_t)blockIdx.x * chunk16 + (size_t)(tid >> 6) * per_wave;
        for (int q = tid & 63; q < per_wave && base + q < n16; q += 64) st16<NT>(dst, base + q, v);
    } else {
        const size_t g = blockIdx.x / grp, w = blockIdx.x % grp;
        const size_t gbase = g * (size_t)grp * chunk16;
        const int piece16 = pace > 1 ? pace : 256;                          // MODE 2: `pace` > 1 carries the piece size (x 16 B)
        const int pieces = chunk16 / piece16;                               // pieces per workgroup
        for (int k = 0; k < pieces; ++k) {
            const size_t pb = gbase + ((size_t)k * grp + w) * piece16;
            for (int r = tid; r < piece16; r += 256)
                if (pb + r < n16) st16<NT>(dst, pb + r, v);
        }
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
};

static size_t lds_for_per_cu(int k) {
    if (k <= 0) return 0;
    const size_t cap = 160 * 1024;
    return ((cap / k + cap / (k + 1)) / 2) & ~(size_t)255;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7, iters = argc > 2 ? atoi(argv[2]) : 20;
    const bool only_wave = argc > 3 && std::string(argv[3]) == "wave";       // wave-autonomous pieces only
    const bool only_rot = argc > 3 && std::string(argv[3]) == "rotate";      // the chunk-size x rotation study only
    const size_t bytes = (size_t)(1 << 20) * 500, n16 = bytes / 16;
    float* d;
    int* dep;
    CK(hipMalloc(&d, bytes + (1 << 20)));
    CK(hipMalloc(&dep, (size_t)(1 << 18) * 64 * 4));
    CK(hipMemset(dep, 0, (size_t)(1 << 18) * 64 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<Variant> vs;
    auto add = [&](const char* tag, int mode, bool nt, int chunk_bytes, int per_cu, int grp, int pace, bool with_dep) {
        char nm[160];
        snprintf(nm, sizeof nm, "%-10s chunk %6d B  NT=%d  wg/CU=%d grp=%d pace=%d dep=%d", tag, chunk_bytes, (int)nt, per_cu, grp, pace, (int)with_dep);
        const int chunk16 = chunk_bytes / 16;
        const unsigned blocks = (unsigned)((n16 + chunk16 - 1) / chunk16);
        const size_t lds = lds_for_per_cu(per_cu);
        const int* dp = with_dep ? dep : nullptr;
        vs.push_back({nm, [=](hipStream_t st) {
            if (mode == 0) { if (nt) hipLaunchKernelGGL((chunk_kernel<true, 0>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp);
                             else hipLaunchKernelGGL((chunk_kernel<false, 0>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp); }
            else if (mode == 1) { if (nt) hipLaunchKernelGGL((chunk_kernel<true, 1>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp);
                                  else hipLaunchKernelGGL((chunk_kernel<false, 1>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp); }
            else if (mode == 5) {
                const int wpb = pace > 0 ? pace : 4;                      // `pace` carries the waves per workgroup
                const size_t pieces = (n16 + chunk16 - 1) / chunk16, waves = (pieces + grp - 1) / grp;
                const unsigned blk = (unsigned)((waves + wpb - 1) / wpb);
                if (nt) hipLaunchKernelGGL((chunk_kernel<true, 5>), dim3(blk), dim3(64 * wpb), lds, st, d, chunk16, n16, grp, 0, dp);
                else hipLaunchKernelGGL((chunk_kernel<false, 5>), dim3(blk), dim3(64 * wpb), lds, st, d, chunk16, n16, grp, 0, dp); }
            else if (mode == 3) { if (nt) hipLaunchKernelGGL((chunk_kernel<true, 3>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp);
                                  else hipLaunchKernelGGL((chunk_kernel<false, 3>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp); }
            else { if (nt) hipLaunchKernelGGL((chunk_kernel<true, 2>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp);
                   else hipLaunchKernelGGL((chunk_kernel<false, 2>), dim3(blocks), dim3(256), lds, st, d, chunk16, n16, grp, pace, dp); }
        }, {}});
    };
    for (int nt = 0; nt < 2; ++nt) {
        add("stripe", 0, nt, 4096, 0, 1, 0, false);                 // one store per thread: the fill
        if (!only_rot && !only_wave)
        for (int cb : {8192, 16384, 32768, 65536})
            for (int cu : {0, 3, 5}) add("stripe", 0, nt, cb, cu, 1, 0, false);
        if (only_wave) {
            for (int pb : {1024, 1936, 3872, 4096, 7744})
                for (int wpb : {1, 4})
                    for (int m : {1, 2, 4})
                        for (int cu : {0, 3, 5}) {
                            if (wpb == 1 && cu) continue;
                            add("wave-piece", 5, nt, pb, cu, m, wpb, true);
                        }
            continue;
        }
        if (only_rot) {
            for (int cb : {15488, 16384, 18432, 30976, 32768, 36864, 16000, 32000, 12800, 25600}) {
                add("stripe", 0, nt, cb, 0, 1, 0, false);
                add("stripe", 0, nt, cb, 3, 1, 0, false);
                for (int mul : {1, 3, 5}) add("rotate", 3, nt, cb, 0, mul, 0, false);
                add("rotate", 3, nt, cb, 3, 1, 0, false);
            }
            continue;
        }
        add("wave-own", 1, nt, 32768, 0, 1, 0, false);
        add("wave-own", 1, nt, 32768, 3, 1, 0, false);
        for (int grp : {8, 64, 256}) add("interleave", 2, nt, 32768, 0, grp, 0, false);
        add("interleave", 2, nt, 32768, 3, 64, 0, false);
        // env-aligned pieces: v2 (500 B per env) 8 / 16 / 32 envs, v1 (400 B) 8 / 16 / 32, C3 (484 B) 16 / 32 envs
        for (int pb : {4000, 8000, 16000, 3200, 6400, 12800, 7744, 15488, 22400})
            for (int grp : {4, 16})
                add("il-piece", 2, nt, pb * 4, 0, grp, pb / 16, false);
        add("il-piece", 2, nt, 16000 * 8, 0, 8, 1000, false);
        add("il-piece", 2, nt, 16000 * 2, 0, 8, 1000, false);
        add("il-piece", 2, nt, 16000 * 4, 3, 8, 1000, false);
        add("il-piece", 2, nt, 16000 * 4, 0, 8, 1000, true);
        add("stripe", 0, nt, 32768, 0, 1, 1, false);                 // paced
        add("stripe", 0, nt, 4096, 0, 1, 0, true);                   // a dependent load first
        add("stripe", 0, nt, 32768, 0, 1, 0, true);
        add("stripe", 0, nt, 32768, 3, 1, 0, true);
        add("interleave", 2, nt, 32768, 0, 64, 0, true);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 10; ++w) for (auto& v : vs) v.launch(s);
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
    printf("buffer %.1f MB\n", bytes / 1e6);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-72s median %8.2f us  min %8.2f us  %7.1f GB/s\n", v.name.c_str(), med * 1e3, v.ms[0] * 1e3, bytes / (med * 1e-3) / 1e9);
    }
    return 0;
}
