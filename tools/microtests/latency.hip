// Developer microbenchmark (GPU box): dependent-chain latencies seen by ONE resident wave on gfx950 -- scalar cache (s_load), LDS (ds_read),
// a transcendental (v_rcp_f32) and a plain dependent VALU op, in shader-clock cycles per link (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency tools/microtests/latency.hip && /tmp/latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
__global__ void k_smem(const int *__restrict__ p, int n, long long *out) {
    int idx = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) idx = __builtin_amdgcn_readfirstlane(p[idx]);
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_lds(const int *__restrict__ p, int n, long long *out) {
    __shared__ int s[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) s[i] = p[i];
    __syncthreads();
    int idx = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) idx = s[idx];
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (idx == -1) out[1] = idx;
}
__global__ void k_rcp(float x, int n, long long *out) {
    float v = x + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) v = __builtin_amdgcn_rcpf(v) + 1.0f;
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (v == -1.f) out[1] = 1;
}
__global__ void k_fma(float x, int n, long long *out) {
    float v = x + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) { v = __builtin_fmaf(v, 1.0001f, 0.5f); v = __builtin_fmaf(v, 0.9999f, -0.5f); }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (v == -1.f) out[1] = 1;
}
__global__ void k_fma4(float x, int n, long long *out) { // four independent chains: 8 v_fma per iteration
    float a = x + threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
        a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 1.0001f, 0.5f); c = __builtin_fmaf(c, 1.0001f, 0.5f); d = __builtin_fmaf(d, 1.0001f, 0.5f);
        a = __builtin_fmaf(a, 0.9999f, -0.5f); b = __builtin_fmaf(b, 0.9999f, -0.5f); c = __builtin_fmaf(c, 0.9999f, -0.5f); d = __builtin_fmaf(d, 0.9999f, -0.5f);
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d == -1.f) out[1] = 1;
}
__global__ void k_mix(float x, int y, int n, long long *out) { // one VALU chain and one SALU chain side by side
    float a = x + threadIdx.x; int v = __builtin_amdgcn_readfirstlane(y);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) { a = __builtin_fmaf(a, 1.0001f, 0.5f); v = v * 3 + 1; a = __builtin_fmaf(a, 0.9999f, -0.5f); v = v ^ (v >> 3); }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
    if (a == -1.f) out[1] = 1;
}
__global__ void k_salu(int x, int n, long long *out) { // dependent scalar ALU chain
    int v = __builtin_amdgcn_readfirstlane(x);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
    for (int i = 0; i < n; ++i) { v = v * 3 + 1; v = v ^ (v >> 3); }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
int main() {
    std::vector<int> h(N);
    for (int i = 0; i < N; ++i) h[i] = (i * 67 + 13) % N; // a long cycle, stays in cache
    int *d; long long *o, ho[2];
    hipMalloc(&d, N * 4); hipMalloc(&o, 16); hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    const int n = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_smem, dim3(1), dim3(64), 0, 0, d, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("s_load chain      : %.1f cycles / link\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("ds_read chain     : %.1f cycles / link\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_rcp, dim3(1), dim3(64), 0, 0, 1.5f, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("v_rcp + v_add     : %.1f cycles / link\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, 1.5f, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("2 dependent v_fma : %.1f cycles / link\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_salu, dim3(1), dim3(64), 0, 0, 7, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("4 dependent s_alu : %.1f cycles / link\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_fma4, dim3(1), dim3(64), 0, 0, 1.5f, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("8 v_fma, 4 chains : %.1f cycles / iteration\n", (double)ho[0] / n);
        hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, 1.5f, 7, n, o); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        if (rep) printf("2 v_fma + 4 s_alu : %.1f cycles / iteration\n", (double)ho[0] / n);
    }
    // clock rate of the counter: time a long loop against wall clock
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, 1.5f, 2000000, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
    printf("counter rate      : %.1f MHz (%.0f counts in %.3f ms)\n", ho[0] / (ms * 1e3), (double)ho[0], ms);
    return 0;
}
