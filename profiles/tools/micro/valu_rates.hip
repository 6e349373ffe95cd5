// Microbenchmark: issue rate of the vector instructions PolyExp is made of, in cycles per wave64 instruction per SIMD
// (4 waves per SIMD, long dependent-free chains).  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8];
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { a[j] = seed + j + threadIdx.x; d[j] = (double)a[j]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) a[j] = __builtin_fmaf(a[j], 1.0001f, 0.5f);            // v_fma_f32
            if (OP == 1) d[j] = d[j] + 1.25;                                     // v_add_f64
            if (OP == 2) d[j] = __builtin_fma(d[j], 1.0001, 0.5);                // v_fma_f64
            if (OP == 3) { d[j] = (double)a[j]; asm volatile("" : "+v"(d[j])); a[j] += 1.0f; }   // v_cvt_f64_f32 (+ one v_add_f32)
            if (OP == 4) { a[j] = (float)d[j]; asm volatile("" : "+v"(a[j])); d[j] = d[j] + 1.0; }  // v_cvt_f32_f64 (+ one v_add_f64)
            if (OP == 5) d[j] = d[j] * 1.0001;                                   // v_mul_f64
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j] + (float)d[j];
    if (s == 12345.f) out[0] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * 4, iters = 20000;   // 4 workgroups x 4 waves = 4 waves per SIMD
    const double clk = p.clockRate * 1e3;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *names[] = {"v_fma_f32", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32 + v_add_f32", "v_cvt_f32_f64 + v_add_f64", "v_mul_f64"};
    for (int op = 0; op < 6; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
                default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            // per SIMD: 4 waves x iters x 8 statements
            if (rep) printf("%-28s %.2f cycles per wave64 statement per SIMD\n", names[op], ms * 1e-3 * clk / (4.0 * iters * 8));
        }
    }
    return 0;
}
