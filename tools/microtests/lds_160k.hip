#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 1) void k(float *out, int nfloats) {
    extern __shared__ float smem[];
    for (int i = threadIdx.x; i < nfloats; i += 256) smem[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    int bad = 0;
    for (int i = threadIdx.x; i < nfloats; i += 256) { if (smem[i] != (float)i) ++bad; acc += smem[i]; }
    out[threadIdx.x] = (float)bad;
}
int main() {
    float *d; hipMalloc(&d, 1024);
    for (int kb : {32, 64, 100, 147, 160}) {
        int nf = kb * 1024 / 4;
        hipError_t r = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, nf * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), nf * 4, 0, d, nf);
        hipError_t r2 = hipDeviceSynchronize();
        float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        float s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%d KB: attr=%s sync=%s last=%s bad=%g\n", kb, hipGetErrorString(r), hipGetErrorString(r2), hipGetErrorString(hipGetLastError()), s);
    }
    return 0;
}
