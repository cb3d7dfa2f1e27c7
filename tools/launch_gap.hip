// Measures the cost of a dependent kernel boundary on one stream (tools only).
// hipcc --offload-arch=gfx950 -O2 tools/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_trivial(float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_rows(float* rows, const float* prev, int nb) {   // like a tracker pass: read all rows, write one
    __shared__ float s[32];
    float a = 0.f;
    for (int b = threadIdx.x >> 5; b < nb; b += blockDim.x >> 5) a += prev[b * 32 + (threadIdx.x & 31)];
    if (threadIdx.x < 32) s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < 32) rows[blockIdx.x * 32 + threadIdx.x] = s[threadIdx.x] * 1e-3f;
}
int main() {
    float *p, *r0, *r1;
    hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    hipMalloc(&r0, 1024 * 128); hipMalloc(&r1, 1024 * 128); hipMemset(r0, 0, 1024 * 128); hipMemset(r1, 0, 1024 * 128);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int wg = cfg == 0 ? 1 : 256, thr = cfg == 3 ? 512 : (cfg == 0 ? 64 : 256), N = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (cfg < 2) hipLaunchKernelGGL(k_trivial, dim3(wg), dim3(thr), 0, st, p, i);
                else hipLaunchKernelGGL(k_rows, dim3(wg), dim3(thr), 0, st, (i & 1) ? r1 : r0, (i & 1) ? r0 : r1, wg);
            }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("cfg %d (%d WG x %d thr, %s): enqueue %.2f us/launch, total %.2f us/launch\n", cfg, wg, thr, cfg < 2 ? "trivial" : "rows",
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    }
    // GPU-side boundary: the same chains replayed from a captured graph (no host launch cost between kernels)
    for (int cfg = 1; cfg < 4; ++cfg) {
        const int wg = 256, thr = cfg == 3 ? 512 : 256, N = 500;
        hipGraph_t graph; hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) {
            if (cfg < 2) hipLaunchKernelGGL(k_trivial, dim3(wg), dim3(thr), 0, st, p, i);
            else hipLaunchKernelGGL(k_rows, dim3(wg), dim3(thr), 0, st, (i & 1) ? r1 : r0, (i & 1) ? r0 : r1, wg);
        }
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            hipGraphLaunch(exec, st);
            hipStreamSynchronize(st);
            auto t2 = std::chrono::steady_clock::now();
            if (rep == 2) printf("graph cfg %d (%d WG x %d thr, %s): %.2f us/kernel\n", cfg, wg, thr, cfg < 2 ? "trivial" : "rows",
                                 std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    }
    return 0;
}
