// Microbenchmarks of the atomic primitives the fusion kernel can be built from (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/atomics_bench.hip -o gpurun_out/atomics_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

struct slot { unsigned long long key; float v[5]; uint32_t aux; };

template <int MODE>
__global__ __launch_bounds__(256) void k_global(slot* tab, uint32_t mask, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    slot* p = tab + (mix(i) & mask);
    const float x = 1.0f + (i & 3);
    if (MODE == 0) { for (int j = 0; j < 5; ++j) unsafeAtomicAdd(&p->v[j], x); }
    if (MODE == 1) { unsafeAtomicAdd(&p->v[0], x); }
    if (MODE == 2) { atomicAdd(&p->key, (unsigned long long)i); }
    if (MODE == 3) { atomicAdd(&p->key, (unsigned long long)i); atomicAdd((unsigned long long*)&p->v[0], 5ull); atomicAdd((unsigned int*)&p->v[2], 7u); }
    if (MODE == 4) { float4 a = *(float4*)&p->v[0]; a.x += x; a.y += x; a.z += x; a.w += x; *(float4*)&p->v[0] = a; }
    if (MODE == 5) { for (int j = 0; j < 5; ++j) __hip_atomic_fetch_add(&p->v[j], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if (MODE == 6) { unsigned long long k = __hip_atomic_load(&p->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (k == 12345) p->aux = 1; }
    if (MODE == 7) { unsigned long long k = atomicCAS(&p->key, 0xFFFFFFFFFFFFFFFFull, (unsigned long long)i); if (k == 12345) p->aux = 1; }
    if (MODE == 8) { for (int j = 0; j < 5; ++j) atomicAdd((unsigned int*)&p->v[j], 3u); }
    if (MODE == 9) { unsigned long long k = p->key; if (k == 12345) p->aux = 1; }
    if (MODE == 10) { for (int j = 0; j < 5; ++j) unsafeAtomicAdd(&tab[(mix(i * 5 + j) & mask)].v[j], x); }   // 5 different lines
}

// LDS: MODE 0 ds_add_f32 distinct addresses, 1 same address per 4 lanes, 2 ds_add_u32 distinct, 3 cmpst_rtn_b64 distinct
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
    __shared__ float lf[4096];
    __shared__ unsigned long long lk[2048];
    for (int i = threadIdx.x; i < 4096; i += 256) lf[i] = 0.f;
    for (int i = threadIdx.x; i < 2048; i += 256) lk[i] = ~0ull;
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t h = mix(t + blockIdx.x * 977);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        if (MODE == 0) { atomicAdd(&lf[(t + it * 67) & 4095], 1.0f); }
        if (MODE == 1) { atomicAdd(&lf[((t >> 2) + it * 67) & 4095], 1.0f); }
        if (MODE == 2) { atomicAdd((unsigned int*)&lf[(t + it * 67) & 4095], 1u); }
        if (MODE == 3) { unsigned long long o = atomicCAS(&lk[(t + it * 67) & 2047], ~0ull, (unsigned long long)t); acc += (float)(o & 1); }
        if (MODE == 4) { atomicAdd(&lf[(h >> 8) & 4095], 1.0f); }                         // random addresses
        if (MODE == 5) { unsigned long long k = *(volatile unsigned long long*)&lk[(h >> 8) & 2047]; acc += (float)(k & 1); }
        if (MODE == 6) { atomicAdd(&lf[((t >> 3) + it * 67) & 4095], 1.0f); }            // 8 lanes per address
        if (MODE == 7) { atomicAdd(&lk[(t + it * 67) & 2047], 3ull); }                    // ds_add_u64 distinct
        if (MODE == 8) { atomicAdd(&lk[((t >> 1) + it * 67) & 2047], 3ull); }             // ds_add_u64 2 lanes/addr
        if (MODE == 9) { atomicAdd(&lk[(h >> 8) & 2047], 3ull); }                         // ds_add_u64 random
        if (MODE == 10) { atomicAdd((unsigned int*)&lf[((t >> 2) + it * 67) & 4095], 1u); } // ds_add_u32 4 lanes/addr
        if (MODE == 11) { atomicAdd(&lk[((t >> 2) + it * 67) & 2047], 3ull); }            // ds_add_u64 4 lanes/addr
        if (MODE == 12) { unsafeAtomicAdd(&lf[(t + it * 67) & 4095], 1.0f); }             // hardware ds_add_f32 distinct
        if (MODE == 13) { unsafeAtomicAdd(&lf[((t >> 2) + it * 67) & 4095], 1.0f); }      // hardware ds_add_f32 4 lanes/addr
        if (MODE == 14) { unsafeAtomicAdd(&lf[(h >> 8) & 4095], 1.0f); }                  // hardware ds_add_f32 random
        if (MODE == 15) { unsafeAtomicAdd((double*)&lk[(t + it * 67) & 2047], 1.0); }     // hardware ds_add_f64 distinct
        if (MODE == 16) { unsafeAtomicAdd((double*)&lk[((t >> 2) + it * 67) & 2047], 1.0); }   // ds_add_f64 4 lanes/addr
        if (MODE == 17) { unsafeAtomicAdd((double*)&lk[(h >> 8) & 2047], 1.0); }          // ds_add_f64 random
    }
    __syncthreads();
    if (t == 0) out[blockIdx.x] = lf[5] + acc;
}

int main() {
    const size_t n_slots = 1u << 22;
    slot* tab; CK(hipMalloc(&tab, n_slots * sizeof(slot)));
    CK(hipMemset(tab, 0, n_slots * sizeof(slot)));
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 1 << 20;
    const char* names[] = { "5x f32 atomic add, same slot", "1x f32 atomic add", "1x u64 atomic add", "2x u64 + 1x u32 atomic add",
        "plain 16B RMW", "5x f32 atomic add wg-scope", "agent atomic load 8B", "CAS 8B (returning)", "5x u32 atomic add same slot",
        "plain load 8B", "5x f32 atomic add, 5 random lines" };
#define RUNG(M) { for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_global<M>, dim3(n / 256), dim3(256), 0, 0, tab, (uint32_t)(n_slots - 1), n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r == 2) printf("global mode %2d %-36s : %8.1f us for %d threads -> %.2f G thread-ops/s\n", M, names[M], ms * 1e3, n, n / ms * 1e-6); } }
    RUNG(0) RUNG(1) RUNG(2) RUNG(3) RUNG(4) RUNG(5) RUNG(6) RUNG(7) RUNG(8) RUNG(9) RUNG(10)
    // same but with a table small enough to sit in L2 (4 MiB / XCD): 64K slots = 2 MiB
#define RUNS(M) { for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_global<M>, dim3(n / 256), dim3(256), 0, 0, tab, (uint32_t)(65536 - 1), n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r == 2) printf("small-table mode %2d %-30s : %8.1f us\n", M, names[M], ms * 1e3); } }
    RUNS(0) RUNS(1) RUNS(4) RUNS(6)
    const char* lnames[] = { "ds_add_f32 distinct", "ds_add_f32 4 lanes/addr", "ds_add_u32 distinct", "ds_cmpst_rtn_b64 distinct", "ds_add_f32 random", "ds_read_b64 random (dependent use)", "ds_add_f32 8 lanes/addr", "ds_add_u64 distinct", "ds_add_u64 2 lanes/addr", "ds_add_u64 random", "ds_add_u32 4 lanes/addr", "ds_add_u64 4 lanes/addr",
        "unsafe ds_add_f32 distinct", "unsafe ds_add_f32 4 lanes/addr", "unsafe ds_add_f32 random", "unsafe ds_add_f64 distinct", "unsafe ds_add_f64 4 lanes/addr", "unsafe ds_add_f64 random" };
    const int iters = 2000, blocks = 256 * 4;
#define RUNL(M) { for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds<M>, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r == 2) { double waveinstr_per_cu = (double)blocks * 4 * iters / 256; printf("lds mode %d %-36s : %8.1f us -> %.1f cycles@2.4GHz per wave-instr per CU\n", M, lnames[M], ms * 1e3, ms * 1e-3 * 2.4e9 / waveinstr_per_cu); } } }
    RUNL(0) RUNL(1) RUNL(2) RUNL(3) RUNL(4) RUNL(5) RUNL(6) RUNL(7) RUNL(8) RUNL(9) RUNL(10) RUNL(11) RUNL(12) RUNL(13) RUNL(14) RUNL(15) RUNL(16) RUNL(17)
    return 0;
}
