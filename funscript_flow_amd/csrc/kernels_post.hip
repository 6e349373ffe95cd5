// gfx950 kernels for the reference's numpy post path on a resident flow field:
//   pass 1: max_divergence (FunscriptFlow.pyw:748-758) + mean flow magnitude (FF:889-890)
//   pass 2: radial_motion_weighted (FF:761-785)
// Both are single-read streaming reductions over the (h, w, 2) float flow: HBM-bound, 8 B per pixel.
// Reductions are wave-shuffle -> LDS -> one partial per workgroup -> a small second kernel, so sums
// are reproducible run to run (no float atomics).
#include "ffl_kernels.h"

#define P1_THREADS 256

int ffl_pass1_blocks(int w, int h) {
    long n = (long)w * h;
    long blocks = (n + P1_THREADS * 8 - 1) / (P1_THREADS * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    return (int)blocks;
}

// np.gradient along one axis: central difference /2 inside, one-sided at the ends (FF:754)
__device__ __forceinline__ float ffl_grad(float lo, float hi, int idx, int n) {
    return (idx == 0 || idx == n - 1) ? hi - lo : (hi - lo) / 2.0f;
}

__device__ __forceinline__ float ffl_div_at(const float2 *__restrict__ flow, int w, int h, int x, int y) {
    int ya = y == 0 ? 0 : y - 1, yb = y == h - 1 ? h - 1 : y + 1;
    int xa = x == 0 ? 0 : x - 1, xb = x == w - 1 ? w - 1 : x + 1;
    float du = ffl_grad(flow[(size_t)ya * w + x].x, flow[(size_t)yb * w + x].x, y, h);   // d(u)/dy
    float dv = ffl_grad(flow[(size_t)y * w + xa].y, flow[(size_t)y * w + xb].y, x, w);   // d(v)/dx
    return du + dv;
}

__device__ __forceinline__ unsigned long long ffl_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double ffl_wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// key = (bits(|div|) << 32) | (0xFFFFFFFF - flat index): the maximum key is the largest |div| and,
// among equals, the smallest row-major index -- np.argmax's first-occurrence rule, order independent.
__global__ __launch_bounds__(P1_THREADS) void k_pass1(PairTab pt, int w, int h, int pov_mode,
                                                      unsigned long long *__restrict__ pkey,
                                                      double *__restrict__ psum) {
    __shared__ unsigned long long skey[P1_THREADS / 64];
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.y;
    const float2 *flow = reinterpret_cast<const float2 *>(pt.flow[b]);
    const unsigned n = (unsigned)w * (unsigned)h;
    unsigned long long key = 0;
    double sum = 0.0;
    // unrolled x4 (same element order per lane): the loads of 4 grid-stride steps are in flight
    // together instead of one memory latency per step
#pragma unroll 4
    for (unsigned i = blockIdx.x * P1_THREADS + threadIdx.x; i < n; i += gridDim.x * P1_THREADS) {
        int y = i / w, x = i - y * w;
        float2 f = flow[i];
        sum += (double)sqrtf(f.x * f.x + f.y * f.y);
        if (!pov_mode) {
            float d = fabsf(ffl_div_at(flow, w, h, x, y));
            unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - i);
            key = k > key ? k : key;
        }
    }
    key = ffl_wave_max_u64(key);
    sum = ffl_wave_sum_f64(sum);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { skey[wv] = key; ssum[wv] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < P1_THREADS / 64; i++) {
            key = skey[i] > key ? skey[i] : key;
            sum += ssum[i];
        }
        pkey[(size_t)b * gridDim.x + blockIdx.x] = key;
        psum[(size_t)b * gridDim.x + blockIdx.x] = sum;
    }
}

__global__ __launch_bounds__(P1_THREADS) void k_pass1_final(PairTab pt, int w, int h, int pov_mode, int nblk,
                                                            const unsigned long long *__restrict__ pkey,
                                                            const double *__restrict__ psum,
                                                            ResTab results) {
    __shared__ unsigned long long skey[P1_THREADS / 64];
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.x;
    unsigned long long key = 0;
    double sum = 0.0;
    for (int i = threadIdx.x; i < nblk; i += P1_THREADS) {
        unsigned long long k = pkey[(size_t)b * nblk + i];
        key = k > key ? k : key;
        sum += psum[(size_t)b * nblk + i];
    }
    key = ffl_wave_max_u64(key);
    sum = ffl_wave_sum_f64(sum);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { skey[wv] = key; ssum[wv] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < P1_THREADS / 64; i++) {
            key = skey[i] > key ? skey[i] : key;
            sum += ssum[i];
        }
        Pass1Result r;
        if (pov_mode) {  // FF:880-882: centre of the bottom edge, value 0
            r.x = w / 2;
            r.y = h - 1;
            r.div_val = 0.f;
        } else {
            unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            r.y = idx / w;
            r.x = idx - r.y * w;
            r.div_val = ffl_div_at(reinterpret_cast<const float2 *>(pt.flow[b]), w, h, r.x, r.y);
        }
        r.pad = 0.f;
        r.mag_sum = sum;
        *results.r[b] = r;
    }
}

void ffl_launch_pass1(PairTab pt, int nB, int w, int h, int pov_mode, unsigned long long *pkey, double *psum,
                      ResTab results, hipStream_t st) {
    int nblk = ffl_pass1_blocks(w, h);
    hipLaunchKernelGGL(k_pass1, dim3(nblk, nB), dim3(P1_THREADS), 0, st, pt, w, h, pov_mode, pkey, psum);
    hipLaunchKernelGGL(k_pass1_final, dim3(nB), dim3(P1_THREADS), 0, st, pt, w, h, pov_mode, nblk, pkey, psum, results);
}

// ---- pass 2: radial_motion_weighted, float64 ------------------------------------------------------
__global__ __launch_bounds__(P1_THREADS) void k_radial(RadialTab rt, int w, int h, int pov_mode,
                                                       double *__restrict__ psum) {
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.y;
    const float2 *flow = reinterpret_cast<const float2 *>(rt.flow[b]);
    const double cx = rt.cx[b], cy = rt.cy[b];
    const double dw = (double)w, dh = (double)h;
    const unsigned n = (unsigned)w * (unsigned)h;
    double sum = 0.0;
#pragma unroll 4
    for (unsigned i = blockIdx.x * P1_THREADS + threadIdx.x; i < n; i += gridDim.x * P1_THREADS) {
        int y = i / w, x = i - y * w;
        float2 f = flow[i];
        double dx = (double)x - cx, dy = (double)y - cy;
        double dot = (double)f.x * dx + (double)f.y * dy;
        if (!pov_mode) {
            dot = ((double)x > cx) ? dot * (double)(w - x) / dw : dot * (double)x / dw;
            dot = ((double)y > cy) ? dot * (double)(h - y) / dh : dot * (double)y / dh;
        }
        sum += dot;
    }
    sum = ffl_wave_sum_f64(sum);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) ssum[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < P1_THREADS / 64; i++) sum += ssum[i];
        psum[(size_t)b * gridDim.x + blockIdx.x] = sum;
    }
}

__global__ __launch_bounds__(P1_THREADS) void k_radial_final(int w, int h, int nblk, const double *__restrict__ psum,
                                                             double *__restrict__ out) {
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.x;
    double sum = 0.0;
    for (int i = threadIdx.x; i < nblk; i += P1_THREADS) sum += psum[(size_t)b * nblk + i];
    sum = ffl_wave_sum_f64(sum);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) ssum[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < P1_THREADS / 64; i++) sum += ssum[i];
        out[b] = sum / ((double)w * (double)h);
    }
}

void ffl_launch_radial(RadialTab rt, int nB, int w, int h, int pov_mode, double *psum, double *out, hipStream_t st) {
    int nblk = ffl_pass1_blocks(w, h);
    hipLaunchKernelGGL(k_radial, dim3(nblk, nB), dim3(P1_THREADS), 0, st, rt, w, h, pov_mode, psum);
    hipLaunchKernelGGL(k_radial_final, dim3(nB), dim3(P1_THREADS), 0, st, w, h, nblk, psum, out);
}
