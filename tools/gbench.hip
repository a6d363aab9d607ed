// gbench.hip -- developer tool (not part of the product or the tests): what a 5x5-window access to a per-env
// float map costs on MI355X as a function of the map's LAYOUT.  Round 3 re-designs the visit map of v4-v6 so that a
// step touches only the window (clock-relative values); the open question this answers is which tile
// shape makes the window cheapest at 1 M envs (1.3-1.7 GB of maps: nothing stays in the Infinity Cache).
//   layouts:  row   reference layout float[G][G] (72-byte rows): a window = 5 segments of 20 B
//             t32   tiles of 2 rows x 4 cells (32 B):   a window = 3 x 2 tiles
//             t64   tiles of 4 x 4 cells (64 B):        a window = 2 x 2 tiles
//             t128  tiles of 4 rows x 8 cells (128 B):  a window = 2 x (1 or 2) tiles
//   modes:    r     gather only        rw   gather, add, write the same granules back
//             +obs  a 700-byte-per-env non-temporal write stream beside it (the observation)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gbench.hip -o tools/gbench && ./tools/gbench [envs] [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int G = 18;

// LAYOUT 0 row, 1 t32, 2 t64, 3 t128.  One "item" = one 16-byte (t32: 8-byte pairs are not used; 16 B = 4 cells of a tile
// row) access; ITEMS items per env, the lanes of a workgroup stripe env*ITEMS + item.
template <int LAYOUT> struct Lay;
template <> struct Lay<0> { static constexpr int REC = G * G * 4, ITEMS = 10; };          // 5 rows x (16 B + 4 B): two loads per row
template <> struct Lay<1> { static constexpr int REC = 5 * 9 * 32, ITEMS = 12; };         // 2 x 3 tiles x 2 rows
template <> struct Lay<2> { static constexpr int REC = 5 * 5 * 64, ITEMS = 16; };         // 2 x 2 tiles x 4 rows
template <> struct Lay<3> { static constexpr int REC = 3 * 5 * 128, ITEMS = 16; };        // 2 x 2 half-tiles x 4 rows (a tile = 2 halves)

template <int LAYOUT, bool WRITE, bool OBS>
__global__ __launch_bounds__(256) void window_kernel(float* maps, float* obs, float* sink, int n, uint32_t seed, int epb) {
    using L = Lay<LAYOUT>;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * epb;
    const int nb = min(epb, n - base);
    float acc = 0.f;
    for (int i = tid; i < nb * L::ITEMS; i += 256) {
        const int le = i / L::ITEMS, it = i - le * L::ITEMS;
        const int e = base + le;
        const uint32_t h = hash32((uint32_t)e * 2654435761u + seed);
        const int ox = (int)(h % 14u), oy = (int)((h >> 8) % 14u);     // window = rows ox..ox+4, columns oy..oy+4
        char* rec = reinterpret_cast<char*>(maps) + (size_t)e * L::REC;
        if (LAYOUT == 0) {
            const int row = it >> 1, part = it & 1;
            float* p = reinterpret_cast<float*>(rec) + (ox + row) * G + oy;
            if (part == 0) {
                struct __attribute__((packed, aligned(4))) Q { float v[4]; };
                Q q = *reinterpret_cast<Q*>(p);
                acc += q.v[0] + q.v[3];
                if (WRITE) { q.v[0] += 1.f; q.v[1] += 1.f; q.v[2] += 1.f; q.v[3] += 1.f; *reinterpret_cast<Q*>(p) = q; }
            } else {
                float v = p[4];
                acc += v;
                if (WRITE) p[4] = v + 1.f;
            }
        } else {
            int tx, ty, sub;   // tile row / column index, 16-byte piece inside the tile
            int off;
            if (LAYOUT == 1) {         // tiles 2 rows x 4 cells: 9 tile rows x 5 tile columns
                const int t = it >> 1; sub = it & 1;
                tx = (ox >> 1) + t / 2; ty = (oy >> 2) + (t & 1);
                off = (tx * 5 + ty) * 32 + sub * 16;
            } else if (LAYOUT == 2) {  // tiles 4 x 4: 5 x 5
                const int t = it >> 2; sub = it & 3;
                tx = (ox >> 2) + (t >> 1); ty = (oy >> 2) + (t & 1);
                off = (tx * 5 + ty) * 64 + sub * 16;
            } else {                   // tiles 4 rows x 8 cells (128 B = 4 rows x 32 B): 5 x 3; an item = 16 B = half a tile row
                const int t = it >> 2; sub = it & 3;
                tx = (ox >> 2) + (t >> 1);
                const int c4 = (oy >> 2) + (t & 1);      // 4-cell column group 0..4 -> tile column c4 / 2, half c4 % 2
                ty = c4 >> 1;
                off = (tx * 3 + ty) * 128 + sub * 32 + (c4 & 1) * 16;
            }
            v4f* p = reinterpret_cast<v4f*>(rec + off);
            v4f v = *p;
            acc += v.x + v.w;
            if (WRITE) { v += 1.f; *p = v; }
        }
    }
    if (OBS) {
        v4f* o = reinterpret_cast<v4f*>(obs) + (size_t)base * 175 / 4 * 1;   // 700 B per env = 43.75 x 16 B; epb % 4 == 0
        const v4f t = {acc, 1.f, 0.f, 1.f};
        const int nq = nb * 175 / 4;
        for (int q = tid; q < nq; q += 256) __builtin_nontemporal_store(t, o + q);
    }
    if (acc == 123456.789f) sink[tid] = acc;
}

// t64 tiles again, but ONE CELL PER LANE: 25 dword loads per env (the window), +1, 25 dword stores back (partial-sector
// writes into lines the loads have just brought into L2) -- against the whole-tile version above.  RECS: 25 more lanes per
// env read a dense 112-byte record (the "previous window" record) instead of a second window.
template <bool WRITE, bool OBS, bool RECS>
__global__ __launch_bounds__(256) void cell_kernel(float* maps, float* recs, float* obs, float* sink, int n, uint32_t seed, int epb) {
    const int tid = threadIdx.x;
    const int base = blockIdx.x * epb;
    const int nb = min(epb, n - base);
    constexpr int IPE = RECS ? 50 : 25;
    float acc = 0.f;
    for (int i = tid; i < nb * IPE; i += 256) {
        const int le = i / IPE, it = i - le * IPE;
        const int e = base + le;
        const uint32_t h = hash32((uint32_t)e * 2654435761u + seed);
        const int ox = (int)(h % 14u), oy = (int)((h >> 8) % 14u);
        if (it < 25) {
            const int x = ox + it / 5, y = oy + it % 5;
            float* p = maps + (size_t)e * 400 + ((x >> 2) * 5 + (y >> 2)) * 16 + (x & 3) * 4 + (y & 3);
            const float v = *p;
            acc += v;
            if (WRITE) *p = v + 1.f;
        } else {
            acc += recs[(size_t)e * 28 + it - 25];
        }
    }
    if (OBS) {
        v4f* o = reinterpret_cast<v4f*>(obs) + (size_t)base * 175 / 4;
        const v4f t = {acc, 1.f, 0.f, 1.f};
        const int nq = nb * 175 / 4;
        for (int q = tid; q < nq; q += 256) __builtin_nontemporal_store(t, o + q);
    }
    if (acc == 123456.789f) sink[tid] = acc;
}

static void run_cells(float* maps, float* recs, float* obs, float* sink, int n, int iters, int epb) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = (n + epb - 1) / epb;
    const char* mn[6] = {"r", "rw", "r+obs", "rw+obs", "r+rec+obs", "rw+rec+obs"};
    for (int mode = 0; mode < 6; ++mode) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            for (int i = 0; i < iters; ++i) {
                const uint32_t seed = 77u + 1000u * i;
                if (mode == 0) hipLaunchKernelGGL((cell_kernel<false, false, false>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
                if (mode == 1) hipLaunchKernelGGL((cell_kernel<true, false, false>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
                if (mode == 2) hipLaunchKernelGGL((cell_kernel<false, true, false>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
                if (mode == 3) hipLaunchKernelGGL((cell_kernel<true, true, false>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
                if (mode == 4) hipLaunchKernelGGL((cell_kernel<false, true, true>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
                if (mode == 5) hipLaunchKernelGGL((cell_kernel<true, true, true>), dim3(blocks), dim3(256), 0, 0, maps, recs, obs, sink, n, seed, epb);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("{\"layout\": \"t64cell\", \"mode\": \"%s\", \"epb\": %d, \"us\": %.1f}\n", mn[mode], epb, ms * 1000.0 / iters);
        fflush(stdout);
    }
}

template <int LAYOUT>
static void run(const char* name, float* maps, float* obs, float* sink, int n, int iters, int epb) {
    using L = Lay<LAYOUT>;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = (n + epb - 1) / epb;
    for (int mode = 0; mode < 4; ++mode) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {   // first repetition = warm-up
            CK(hipEventRecord(a));
            for (int i = 0; i < iters; ++i) {
                const uint32_t seed = 77u + 1000u * i;
                if (mode == 0) hipLaunchKernelGGL((window_kernel<LAYOUT, false, false>), dim3(blocks), dim3(256), 0, 0, maps, obs, sink, n, seed, epb);
                if (mode == 1) hipLaunchKernelGGL((window_kernel<LAYOUT, true, false>), dim3(blocks), dim3(256), 0, 0, maps, obs, sink, n, seed, epb);
                if (mode == 2) hipLaunchKernelGGL((window_kernel<LAYOUT, false, true>), dim3(blocks), dim3(256), 0, 0, maps, obs, sink, n, seed, epb);
                if (mode == 3) hipLaunchKernelGGL((window_kernel<LAYOUT, true, true>), dim3(blocks), dim3(256), 0, 0, maps, obs, sink, n, seed, epb);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        const char* mn[4] = {"r", "rw", "r+obs", "rw+obs"};
        const double us = ms * 1000.0 / iters;
        const double gran = (LAYOUT == 0 ? 5 * 20 : L::ITEMS * 16) * (mode & 1 ? 2.0 : 1.0) + (mode >= 2 ? 700.0 : 0.0);
        printf("{\"layout\": \"%s\", \"mode\": \"%s\", \"epb\": %d, \"us\": %.1f, \"bytes_per_env_issued\": %.0f, \"TBps_issued\": %.2f}\n",
               name, mn[mode], epb, us, gran, gran * n / us * 1e-6);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 20;
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    float *maps, *obs, *sink;
    CK(hipMalloc(&maps, (size_t)n * 1920 + 4096));
    CK(hipMalloc(&obs, (size_t)n * 700 + 4096));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(maps, 0, (size_t)n * 1920));
    float* recs;
    CK(hipMalloc(&recs, (size_t)n * 112 + 4096));
    CK(hipMemset(recs, 0, (size_t)n * 112));
    const bool all = argc > 3;
    for (int epb : {32, 64, 128}) {
        if (all) run<0>("row", maps, obs, sink, n, iters, epb);
        if (all) run<1>("t32", maps, obs, sink, n, iters, epb);
        run<2>("t64", maps, obs, sink, n, iters, epb);
        if (all) run<3>("t128", maps, obs, sink, n, iters, epb);
        run_cells(maps, recs, obs, sink, n, iters, epb);
    }
    // the obs stream alone, for reference
    return 0;
}
