// wbench_alloc.hip -- developer tool: does the ALLOCATION behind a write target decide the rate of a many-stream write
// pattern (tools/placement_pmc.py)?  Eight separate hipMalloc'ed buffers and eight sub-ranges of one
// large allocation, each written with (a) the fill pattern, (b) 32-KiB private chunks per workgroup, uncapped.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/wbench_alloc.hip -o tools/wbench_alloc && ./tools/wbench_alloc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void chunk_kernel(float* dst, int chunk16, size_t n16) {
    const size_t base = (size_t)blockIdx.x * chunk16;
    const v4f v = {1.f, 0.f, 1.f, 0.f};
    for (int q = threadIdx.x; q < chunk16 && base + q < n16; q += 256) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst) + base + q);
}

static float time_us(float* d, int chunk_bytes, size_t bytes, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t n16 = bytes / 16;
    const int chunk16 = chunk_bytes / 16;
    const unsigned blocks = (unsigned)((n16 + chunk16 - 1) / chunk16);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(chunk_kernel, dim3(blocks), dim3(256), 0, s, d, chunk16, n16);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(chunk_kernel, dim3(blocks), dim3(256), 0, s, d, chunk16, n16);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3f;
}

int main() {
    const size_t bytes = (size_t)(1 << 20) * 484;        // the C3 observation buffer
    hipStream_t s;
    hipEvent_t e0, e1;
    CK(hipStreamCreate(&s)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float*> bufs;
    for (int i = 0; i < 8; ++i) {
        float* d; void* pad;
        CK(hipMalloc(&d, bytes));
        CK(hipMalloc(&pad, (size_t)(3 + 5 * (i % 4)) << 20));
        bufs.push_back(d);
    }
    float* big;
    CK(hipMalloc(&big, bytes * 8));
    for (int w = 0; w < 200; ++w) time_us(bufs[0], 4096, bytes, s, e0, e1);     // warm the device
    for (int rnd = 0; rnd < 2; ++rnd) {
        for (int i = 0; i < 8; ++i)
            printf("round %d  separate alloc %d (%p)   fill %6.1f us   32-KiB chunks %6.1f us\n", rnd, i, (void*)bufs[i],
                   time_us(bufs[i], 4096, bytes, s, e0, e1), time_us(bufs[i], 32768, bytes, s, e0, e1));
        for (int i = 0; i < 8; ++i) {
            float* d = big + (size_t)i * (bytes / 4);
            printf("round %d  big alloc +%d x 484 MiB        fill %6.1f us   32-KiB chunks %6.1f us\n", rnd, i,
                   time_us(d, 4096, bytes, s, e0, e1), time_us(d, 32768, bytes, s, e0, e1));
        }
    }
    return 0;
}
