// Microbenchmark: what one wave64 global load costs the CU's vector-memory path when the data is cache-resident.
// 12 waves per CU (3 workgroups of 256) each issue ITER x 8 independent loads of 4 / 8 / 16 bytes per lane from a
// 64 KB window (L1/L2 hits), addresses contiguous across lanes (the shape of the UpdateMatrices gathers).
// Prints CU-cycles per wave-instruction = time * clock * CUs / wave-instructions.
// Build: hipcc -O3 --offload-arch=gfx950 vmem_issue.hip -o vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(256) void k(const float *__restrict__ buf, float *out, int iters, unsigned mask) {
    const unsigned lane = threadIdx.x;
    unsigned off = (blockIdx.x * 977u + lane * (BYTES / 4)) & mask;   // float index
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned o = (off + j * 4099u * (BYTES / 4)) & mask & ~(unsigned)(BYTES / 4 - 1);
            if (BYTES == 4) v[j] = buf[o];
            else if (BYTES == 8) { float2 t = *reinterpret_cast<const float2 *>(buf + o); v[j] = t.x + t.y; }
            else { float4 t = *reinterpret_cast<const float4 *>(buf + o); v[j] = t.x + t.y + t.z + t.w; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j];
        off = (off + 64u * (BYTES / 4) + (unsigned)(acc > 1e30f)) & mask;
    }
    if (acc == 123.456f) out[0] = acc;
}

// dword loads whose lanes are 8 bytes apart (the layout of a two-pixels-per-lane kernel reading one of its pixels)
__global__ __launch_bounds__(256) void k_stride2(const float *__restrict__ buf, float *out, int iters, unsigned mask) {
    unsigned off = (blockIdx.x * 977u + threadIdx.x * 2u) & mask;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = buf[(off + j * 4099u * 2u) & mask];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j];
        off = (off + 128u + (unsigned)(acc > 1e30f)) & mask;
    }
    if (acc == 123.456f) out[0] = acc;
}

// stores of 4 / 8 / 16 bytes per lane, contiguous across lanes, into a 64 KB window per workgroup
template <int BYTES>
__global__ __launch_bounds__(256) void ks(float *__restrict__ buf, int iters, unsigned mask) {
    unsigned off = (blockIdx.x * 16384u + threadIdx.x * (BYTES / 4));
    const unsigned base = blockIdx.x * 16384u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned o = base + ((off + j * 1024u) & mask & ~(unsigned)(BYTES / 4 - 1));
            if (BYTES == 4) buf[o] = (float)it;
            else if (BYTES == 8) *reinterpret_cast<float2 *>(buf + o) = make_float2((float)it, 1.f);
            else *reinterpret_cast<float4 *>(buf + o) = make_float4((float)it, 1.f, 2.f, 3.f);
        }
        off += 256u * (BYTES / 4);
    }
}

int main() {
    const int n = 1 << 14;   // 16 K floats = 64 KB
    float *buf, *out;
    hipMalloc(&buf, n * 4 + 64);
    hipMalloc(&out, 4);
    std::vector<float> h(n + 16, 1.0f);
    hipMemcpy(buf, h.data(), n * 4 + 64, hipMemcpyHostToDevice);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * 3, iters = 4000;
    const double clk = p.clockRate * 1e3;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int bytes : {4, 8, 16}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (bytes == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, buf, out, iters, (unsigned)(n - 1));
            if (bytes == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, buf, out, iters, (unsigned)(n - 1));
            if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, buf, out, iters, (unsigned)(n - 1));
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double winstr_per_cu = 12.0 * iters * 8;   // wave-instructions each CU issues
            if (rep) printf("%2d B/lane: %.3f ms, %.1f CU-cycles per wave64 load (clock %.0f MHz, %d CUs), %.1f B/clk/CU\n", bytes, ms,
                            ms * 1e-3 * clk / winstr_per_cu, clk / 1e6, cus, 64.0 * bytes / (ms * 1e-3 * clk / winstr_per_cu));
        }
    }
    {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stride2, dim3(grid), dim3(256), 0, 0, buf, out, iters, (unsigned)(n - 1));
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("dword, lanes 8 B apart: %.1f CU-cycles per wave64 load\n", ms * 1e-3 * clk / (12.0 * iters * 8));
    }
    float *sbuf;
    hipMalloc(&sbuf, (size_t)grid * 16384 * 4);
    for (int bytes : {4, 8, 16}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (bytes == 4) hipLaunchKernelGGL(ks<4>, dim3(grid), dim3(256), 0, 0, sbuf, 1000, 16383u);
            if (bytes == 8) hipLaunchKernelGGL(ks<8>, dim3(grid), dim3(256), 0, 0, sbuf, 1000, 16383u);
            if (bytes == 16) hipLaunchKernelGGL(ks<16>, dim3(grid), dim3(256), 0, 0, sbuf, 1000, 16383u);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("store %2d B/lane: %.1f CU-cycles per wave64 store, %.1f B/clk/CU (write-back bound: 64 KB window per workgroup)\n", bytes,
                            ms * 1e-3 * clk / (12.0 * 1000 * 8), 64.0 * bytes / (ms * 1e-3 * clk / (12.0 * 1000 * 8)));
        }
    }
    return 0;
}
