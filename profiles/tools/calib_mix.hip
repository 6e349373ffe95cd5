// What the memory system sustains for the read : write mixes of this path's kernels (pure streaming, 16 bytes
// per lane, no arithmetic): the practical ceiling to hold the kernels' achieved GB/s against.
//   hipcc --offload-arch=gfx950 -O3 -o calib_mix calib_mix.hip && ./calib_mix
// NR read streams and NW write streams of `n` float4 each (distinct arrays, like the M / R0 / R1 / flow planes).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NR, int NW>
__global__ __launch_bounds__(256) void k_mix(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float4 v = in[(size_t)r * n + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) out[(size_t)w * n + i] = a;
        if (NW == 0 && a.x == 12345.678f) out[0] = a;
    }
}

template <int NR, int NW>
static void run(const char *what, const float4 *in, float4 *out, size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<NR, NW>), dim3(16384), dim3(256), 0, 0, in, out, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2)
            printf("%-44s %d read + %d write streams of %.0f MiB: %.1f us, %.0f GB/s\n", what, NR, NW, n * 16.0 / (1 << 20),
                   ms * 1e3, (NR + NW) * n * 16.0 / (ms * 1e-3) / 1e9);
    }
}

int main() {
    const size_t n = (size_t)16 << 20;  // 16 Mi float4 = 256 MiB per stream
    float4 *in, *out;
    hipMalloc(&in, 15 * n * 16);
    hipMalloc(&out, 7 * n * 16);
    hipMemset(in, 0, 15 * n * 16);
    hipMemset(out, 0, 7 * n * 16);
    run<5, 0>("read only", in, out, n);
    run<0, 5>("write only", in, out, n);
    run<5, 2>("blur+solve (M 20 in, flow 8 out)", in, out, n);
    run<15, 7>("fused blur+solve+update (60 in, 28 out)", in, out, n);
    run<12, 7>("update matrices + upsample (~50 in, 28 out)", in, out, n);
    run<1, 5>("polyexp (4 in, 20 out)", in, out, n);
    run<2, 0>("pass 1 / pass 2 (8 in)", in, out, n);
    return 0;
}
