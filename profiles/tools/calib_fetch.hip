// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the ffl kernels use
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
// Each kernel streams a 1 GiB buffer exactly once with 4, 8 or 16 bytes per lane and writes 256 MiB.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ in, float *__restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T v = in[i];
        const float *p = reinterpret_cast<const float *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += p[k];
    }
    if (acc == 12345.678f) out[0] = acc;  // never true: keeps the loads alive without a store stream
}

template <typename T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ out, size_t n) {
    T v;
    float *p = reinterpret_cast<float *>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) p[k] = 1.0f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

int main() {
    const size_t bytes = 1ull << 30, wbytes = 1ull << 28;
    float *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, wbytes);
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read<float>, dim3(4096), dim3(256), 0, 0, buf, out, bytes / 4);
        hipLaunchKernelGGL(k_read<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, out, bytes / 8);
        hipLaunchKernelGGL(k_read<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, out, bytes / 16);
        hipLaunchKernelGGL(k_write<float>, dim3(4096), dim3(256), 0, 0, out, wbytes / 4);
        hipLaunchKernelGGL(k_write<float2>, dim3(4096), dim3(256), 0, 0, (float2 *)out, wbytes / 8);
        hipLaunchKernelGGL(k_write<float4>, dim3(4096), dim3(256), 0, 0, (float4 *)out, wbytes / 16);
    }
    hipDeviceSynchronize();
    printf("read 1 GiB x {4,8,16} B/lane, wrote 256 MiB x {4,8,16} B/lane, twice\n");
    return 0;
}
