#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int *a, int *o, int nact) {
    if ((int)threadIdx.x >= nact) return;
    int v = a[threadIdx.x];
    int r = __builtin_amdgcn_update_dpp((int)0xFFFFFFFF, v, 0x130, 0xf, 0xf, false);
    o[threadIdx.x] = r;
}
int main() {
    int h[64], *a, *o, r[64];
    for (int i = 0; i < 64; i++) h[i] = 100 + i;
    hipMalloc(&a, 256); hipMalloc(&o, 256);
    hipMemcpy(a, h, 256, hipMemcpyHostToDevice);
    for (int nact : {64, 40}) {
        hipMemset(o, 0, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o, nact);
        hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
        printf("nact=%d:", nact);
        for (int i = 0; i < 64; i++) if (i < 3 || (i >= 14 && i <= 17) || (i >= 30 && i <= 33) || (i >= 38 && i <= 41) || i >= 61) printf(" [%d]=%d", i, r[i]);
        printf("\n");
    }
    return 0;
}
