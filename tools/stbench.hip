// Cache-policy bits of the streaming store: the same one-store-per-thread fill (the pattern of the library's bandwidth
// probe) over a buffer of the C3 observation's size, the 16-byte store issued with each combination of the gfx950
// sc0 / sc1 / nt bits.    hipcc -O3 --offload-arch=gfx950 tools/stbench.hip -o tools/stbench && tools/stbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(int4* dst, size_t n16, int v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i val = {v, v + 1, v + 2, v + 3};
    int4* p = dst + i;
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(val) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(val) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(val) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(val) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(val) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(val) : "memory");
}

template <int MODE>
static float run(int4* dst, size_t n16, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n16 + 255) / 256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dst, n16, i);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, dst, n16, i);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)1048576 * 121 * 4;
    const size_t n16 = bytes / 16;
    int4* dst;
    CHECK(hipMalloc(&dst, bytes));
    const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    for (int pass = 0; pass < 3; ++pass) {
        float ms[8] = {run<0>(dst, n16, 20), run<1>(dst, n16, 20), run<2>(dst, n16, 20), run<3>(dst, n16, 20),
                       run<4>(dst, n16, 20), run<5>(dst, n16, 20), run<6>(dst, n16, 20), run<7>(dst, n16, 20)};
        for (int m = 0; m < 8; ++m)
            printf("{\"pass\": %d, \"bits\": \"%s\", \"bytes\": %zu, \"us\": %.2f, \"GBps\": %.0f}\n", pass, names[m], bytes, ms[m] * 1e3,
                   bytes / (ms[m] * 1e-3) / 1e9);
    }
    return 0;
}
