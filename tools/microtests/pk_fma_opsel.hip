#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define T(NAME, MODS) \
__global__ void NAME(float *out, float a0, float a1) { \
    f2 sp; sp.x = a0; sp.y = a1; \
    f2 v; v.x = 3.f + threadIdx.x; v.y = 5.f + threadIdx.x; \
    f2 c; c.x = 100.f; c.y = 1000.f; \
    f2 vs = sp; f2 r1, r2; \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(r1) : "s"(sp), "v"(v), "v"(c)); \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(r2) : "v"(vs), "v"(v), "v"(c)); \
    if (threadIdx.x == 0) { out[0] = r1.x; out[1] = r1.y; out[2] = r2.x; out[3] = r2.y; } \
}
T(k0, "")
T(k1, "op_sel:[1,0,0]")
T(k2, "op_sel:[0,1,0] op_sel_hi:[0,0,1]")
T(k3, "op_sel:[1,0,0] op_sel_hi:[0,1,1]")
T(k4, "op_sel_hi:[0,1,1]")
T(k5, "op_sel:[1,1,0] op_sel_hi:[0,0,1]")
T(k6, "op_sel:[0,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
T(k7, "op_sel:[1,0,0] op_sel_hi:[1,1,1]")
int main() {
    float *d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
    void (*ks[])(float *, float, float) = {k0, k1, k2, k3, k4, k5, k6, k7};
    const char *names[] = {"plain", "op_sel:[1,0,0]", "op_sel:[0,1,0] hi:[0,0,1]", "op_sel:[1,0,0] hi:[0,1,1]", "hi:[0,1,1]", "op_sel:[1,1,0] hi:[0,0,1]", "neg src0", "op_sel:[1,0,0] hi:[1,1,1]"};
    for (int i = 0; i < 8; ++i) {
        hipLaunchKernelGGL(ks[i], dim3(1), dim3(64), 0, 0, d, 2.f, 7.f);
        float h[4]; if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        printf("%-28s sgpr: %8g %8g   vgpr: %8g %8g   %s\n", names[i], h[0], h[1], h[2], h[3], (h[0] == h[2] && h[1] == h[3]) ? "same" : "DIFFERENT");
    }
    return 0;
}
