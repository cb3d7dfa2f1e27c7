// Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE for the access pattern of k_fuse's flush:
// N distinct 32-byte records of a 128 MB table, scattered, read and written as two 16-byte agent-scope (sc1) accesses per lane.
// Run each mode under `rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE` (tools/write_calib.sh); known bytes = N x 32 each way.
//   mode 0  2 x 16-B sc1 stores per lane, scattered records            (the flush)
//   mode 1  2 x 16-B plain stores per lane, scattered records
//   mode 2  1 x 16-B sc1 store per lane, lane pairs share a record      (32 contiguous bytes per pair)
//   mode 3  2 x 16-B sc1 stores per lane, DENSE records (lane i -> record i): the counter's behaviour on coalesced streams
//   mode 4  2 x 16-B sc1 loads per lane, scattered records              (the flush's read side; result kept alive by a rare store)
//   mode 5  2 x 16-B sc1 loads, dense
// hipcc --offload-arch=gfx950 -O3 tools/write_calib.hip -o tools/bin/write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_calib(char* tab, uint32_t rec_mask, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (MODE == 2) {
        const uint32_t r = i >> 1;
        if (r >= n) return;
        char* p = tab + (size_t)((r * 2654435761u) & rec_mask) * 32 + (i & 1) * 16;
        const u32x4 v = { i, 1u, 2u, 3u };
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        return;
    }
    if (i >= n) return;
    const uint32_t r = (MODE == 3 || MODE == 5) ? i : ((i * 2654435761u) & rec_mask);     // odd multiplier: a bijection on the mask
    char* p = tab + (size_t)r * 32;
    if (MODE == 0 || MODE == 3) {
        const u32x4 v = { i, 1u, 2u, 3u };
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1" :: "v"(p), "v"(v) : "memory");
    } else if (MODE == 1) {
        u32x4* q = reinterpret_cast<u32x4*>(p);
        q[0] = u32x4{ i, 1u, 2u, 3u };
        q[1] = u32x4{ i, 4u, 5u, 6u };
    } else {
        u32x4 a, b;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1" : "=v"(b) : "v"(p) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) :: "memory");
        if ((a.x ^ b.y) == 0xDEADBEEFu) reinterpret_cast<uint32_t*>(tab)[0] = 1u;
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 1260000u;
    const uint32_t n_rec = 1u << 22;                       // 128 MB of 32-byte records: the bench map's table
    char* tab = nullptr;
    CK(hipMalloc((void**)&tab, (size_t)n_rec * 32));
    CK(hipMemset(tab, 0, (size_t)n_rec * 32));
    CK(hipDeviceSynchronize());
    const uint32_t threads = mode == 2 ? 2 * n : n;
    const dim3 grid((threads + 255) / 256), block(256);
    for (int rep = 0; rep < 5; ++rep) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k_calib<0>, grid, block, 0, 0, tab, n_rec - 1, n); break;
            case 1: hipLaunchKernelGGL(k_calib<1>, grid, block, 0, 0, tab, n_rec - 1, n); break;
            case 2: hipLaunchKernelGGL(k_calib<2>, grid, block, 0, 0, tab, n_rec - 1, n); break;
            case 3: hipLaunchKernelGGL(k_calib<3>, grid, block, 0, 0, tab, n_rec - 1, n); break;
            case 4: hipLaunchKernelGGL(k_calib<4>, grid, block, 0, 0, tab, n_rec - 1, n); break;
            default: hipLaunchKernelGGL(k_calib<5>, grid, block, 0, 0, tab, n_rec - 1, n); break;
        }
        CK(hipDeviceSynchronize());
    }
    printf("mode %d: %u records x 32 B = %.2f MB per launch\n", mode, n, n * 32.0 / 1e6);
    return 0;
}
