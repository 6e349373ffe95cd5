// gfx950 kernels for the reference's numpy post path on a resident flow field:
//   pass 1: max_divergence (FunscriptFlow.pyw:748-758) + mean flow magnitude (FF:889-890)
//   pass 2: radial_motion_weighted (FF:761-785)
// Both are single-read streaming reductions over the (h, w, 2) float flow: HBM-bound, 8 B per pixel.
// Reductions are wave-shuffle -> LDS -> one partial per workgroup -> a small second kernel, so sums
// are reproducible run to run (no float atomics).
#include "ffl_kernels.h"

#define P1_THREADS 256
#define P1_RG 16          // rows a wave walks down (its row group)
#ifndef P1_G
#define P1_G 16            // rows of it whose loads are in flight together in k_pass1 (4: 118 us, 8: 111, 16: 105 at 1080p B = 32)
#endif
#define P1_STRIP 126      // useful pixels of a pass-1 strip: 64 lanes x 2 pixels minus one halo pixel per side
#define P2_STRIP 128      // pass 2 needs no halo

// Both passes walk the flow field in column strips: a wave owns 128 consecutive pixels of a row (two per
// lane, one 16-byte load) and walks down P1_RG rows, so every pixel is read once with full-width loads,
// the vertical neighbours of pass 1 are the previous / next row already in registers and the horizontal
// ones come from the adjacent lanes.  No per-pixel index division, no neighbour gathers.
static inline int ffl_strip_waves(int w, int h, int strip) {
    return ((w + strip - 1) / strip) * ((h + P1_RG - 1) / P1_RG);
}
int ffl_pass1_blocks(int w, int h) {  // workgroups (= partial results) per pair, the larger of the two passes
    return (ffl_strip_waves(w, h, P1_STRIP) + 3) / 4;
}

// np.gradient along one axis: central difference /2 inside, one-sided at the ends (FF:754)
__device__ __forceinline__ float ffl_grad(float lo, float hi, int idx, int n) {
    return (idx == 0 || idx == n - 1) ? hi - lo : (hi - lo) / 2.0f;
}

__device__ __forceinline__ float ffl_div_at(const float2 *__restrict__ flow, int w, int h, int x, int y) {
    int ya = y == 0 ? 0 : y - 1, yb = y == h - 1 ? h - 1 : y + 1;
    int xa = x == 0 ? 0 : x - 1, xb = x == w - 1 ? w - 1 : x + 1;
    float du = ffl_grad(ffl_gload2(flow, 8u * ((size_t)ya * w + x)).x, ffl_gload2(flow, 8u * ((size_t)yb * w + x)).x, y, h);   // d(u)/dy
    float dv = ffl_grad(ffl_gload2(flow, 8u * ((size_t)y * w + xa)).y, ffl_gload2(flow, 8u * ((size_t)y * w + xb)).y, x, w);   // d(v)/dx
    return du + dv;
}

// the lane's two pixels (x, x+1) of row y, each clamped into the image, with one 16-byte load
__device__ __forceinline__ void ffl_load_pair(const float2 *__restrict__ flow, int w, int h, int x, int y, float2 &p0,
                                              float2 &p1) {
    const int yc = min(max(y, 0), h - 1);
    const int xa = min(max(x, 0), w - 2);  // pair start inside the row
    const float4 t = ffl_gload4(flow, 8u * ((unsigned)yc * (unsigned)w + (unsigned)xa))  /* < 2^32: ffl_create */;
    const float2 A = make_float2(t.x, t.y), B = make_float2(t.z, t.w);
    p0 = (min(max(x, 0), w - 1) == xa) ? A : B;
    p1 = (min(max(x + 1, 0), w - 1) == xa) ? A : B;
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
__global__ __launch_bounds__(P1_THREADS) void k_pass1(const PairTab *__restrict__ pt, int w, int h, int pov_mode,
                                                      unsigned long long *__restrict__ pkey,
                                                      double *__restrict__ psum) {
    __shared__ unsigned long long skey[P1_THREADS / 64];
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.y;
    const float2 *flow = reinterpret_cast<const float2 *>(pt->flow[0][b]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nstrips = (w + P1_STRIP - 1) / P1_STRIP, ngroups = (h + P1_RG - 1) / P1_RG;
    const int wid = blockIdx.x * (P1_THREADS / 64) + wv;  // wave-uniform
    unsigned long long key = 0;
    double sum = 0.0;
    if (wid < nstrips * ngroups) {
        const int grp = wid / nstrips, strip = wid - grp * nstrips;
        const int x = strip * P1_STRIP - 1 + 2 * lane, y0 = grp * P1_RG;  // the lane's pixels: x, x+1
        // a pixel counts if it is inside the image and not one of the strip's two halo pixels
        const bool ok0 = lane > 0 && x < w, ok1 = lane < 63 && x + 1 < w;
        // Two groups of 8 rows.  All 10 rows a group needs are requested before anything is computed: written as one
        // loop (load row y + 1, use rows y - 1 .. y + 1) the compiler put a full wait behind every load -- the uniform
        // `pov_mode` branch and the lane exchanges sit between them -- and a wave paid 18 dependent round trips.
        const int xa = min(max(x, 0), w - 2);   // the lane's 16-byte window starts here (inside the row)
        const bool first0 = min(max(x, 0), w - 1) == xa, first1 = min(max(x + 1, 0), w - 1) == xa;
#pragma unroll 1
        for (int g = 0; g < P1_RG; g += P1_G) {
            float4 raw[P1_G + 2];
#pragma unroll
            for (int r = 0; r < P1_G + 2; r++) {
                const int yc = min(max(y0 + g - 1 + r, 0), h - 1);   // clamped: rows past the end repeat row h-1
                raw[r] = ffl_gload4(flow, 8u * ((unsigned)yc * (unsigned)w + (unsigned)xa));  // < 2^32: ffl_create
            }
#pragma unroll
            for (int r = 0; r < P1_G; r++) {
                const int y = y0 + g + r;
                const float4 tu = raw[r], tc = raw[r + 1], td = raw[r + 2];
                const float2 up0 = first0 ? make_float2(tu.x, tu.y) : make_float2(tu.z, tu.w), up1 = first1 ? make_float2(tu.x, tu.y) : make_float2(tu.z, tu.w);
                const float2 c0 = first0 ? make_float2(tc.x, tc.y) : make_float2(tc.z, tc.w), c1 = first1 ? make_float2(tc.x, tc.y) : make_float2(tc.z, tc.w);
                const float2 dn0 = first0 ? make_float2(td.x, td.y) : make_float2(td.z, td.w), dn1 = first1 ? make_float2(td.x, td.y) : make_float2(td.z, td.w);
                const bool row_ok = y < h;
                sum += (ok0 && row_ok) ? (double)sqrtf(c0.x * c0.x + c0.y * c0.y) : 0.0;
                sum += (ok1 && row_ok) ? (double)sqrtf(c1.x * c1.x + c1.y * c1.y) : 0.0;
                if (!pov_mode) {
                    // horizontal neighbours: the adjacent lanes' pixels (clamped loads make x = 0 / w-1 see themselves)
                    const float left0 = __shfl_up(c1.y, 1, 64), right1 = __shfl_down(c0.y, 1, 64);
                    const float d0 = fabsf(ffl_grad(up0.x, dn0.x, y, h) + ffl_grad(left0, c1.y, x, w));
                    const float d1 = fabsf(ffl_grad(up1.x, dn1.x, y, h) + ffl_grad(c0.y, right1, x + 1, w));
                    const unsigned i0 = (unsigned)y * (unsigned)w + (unsigned)x;
                    if (ok0 && row_ok) {
                        const unsigned long long k = ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned long long)(0xFFFFFFFFu - i0);
                        key = k > key ? k : key;
                    }
                    if (ok1 && row_ok) {
                        const unsigned long long k = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned long long)(0xFFFFFFFFu - (i0 + 1u));
                        key = k > key ? k : key;
                    }
                }
            }
        }
    }
    key = ffl_wave_max_u64(key);
    sum = ffl_wave_sum_f64(sum);
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

__global__ __launch_bounds__(P1_THREADS) void k_pass1_final(const PairTab *__restrict__ pt, int w, int h, int pov_mode,
                                                            int nblk, const unsigned long long *__restrict__ pkey,
                                                            const double *__restrict__ psum) {
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
            r.div_val = ffl_div_at(reinterpret_cast<const float2 *>(pt->flow[0][b]), w, h, r.x, r.y);
        }
        r.pad = 0.f;
        r.mag_sum = sum;
        *pt->res[b] = r;
    }
}

void ffl_launch_pass1(const PairTab *pt, int nB, int w, int h, int pov_mode, unsigned long long *pkey, double *psum,
                      hipStream_t st) {
    int nblk = (ffl_strip_waves(w, h, P1_STRIP) + 3) / 4;
    hipLaunchKernelGGL(k_pass1, dim3(nblk, nB), dim3(P1_THREADS), 0, st, pt, w, h, pov_mode, pkey, psum);
    hipLaunchKernelGGL(k_pass1_final, dim3(nB), dim3(P1_THREADS), 0, st, pt, w, h, pov_mode, nblk, pkey, psum);
}

// ---- pass 2: radial_motion_weighted, float64 ------------------------------------------------------
// wytab[y] = (double)(h - y) / h, wytab[h + y] = (double)y / h: the two row weights of FF:780-783, formed once per
// context on the host (IEEE division, the value the device's division yields).  The row index is wave-uniform, so the
// weight comes with a scalar load instead of a 15-instruction f64 division per lane and row -- the kernel had
// hoisted all 16 of them and needed 198 VGPRs (2 waves per SIMD).
__global__ __launch_bounds__(P1_THREADS) void k_radial(const RadialTab *__restrict__ rt, int w, int h, int pov_mode,
                                                       const double *__restrict__ wytab, double *__restrict__ psum) {
    __shared__ double ssum[P1_THREADS / 64];
    const int b = blockIdx.y;
    const float2 *flow = reinterpret_cast<const float2 *>(rt->flow[b]);
    const double cx = rt->cx[b], cy = rt->cy[b];
    const double dw = (double)w;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nstrips = (w + P2_STRIP - 1) / P2_STRIP, ngroups = (h + P1_RG - 1) / P1_RG;
    const int wid = blockIdx.x * (P1_THREADS / 64) + wv;  // wave-uniform
    double sum = 0.0;
    if (wid < nstrips * ngroups) {
        const int grp = wid / nstrips, strip = wid - grp * nstrips;
        const int x = strip * P2_STRIP + 2 * lane, y0 = grp * P1_RG;  // the lane's pixels: x, x+1
        const bool ok0 = x < w, ok1 = x + 1 < w;
        // per-column terms once per lane: dx and the quadrant weight (w - x) / w or x / w  (FF:776-779)
        const double dx0 = (double)x - cx, dx1 = (double)(x + 1) - cx;
        const double wx0 = pov_mode ? 1.0 : (((double)x > cx) ? (double)(w - x) / dw : (double)x / dw);
        const double wx1 = pov_mode ? 1.0 : (((double)(x + 1) > cx) ? (double)(w - x - 1) / dw : (double)(x + 1) / dw);
        // two groups of 8 rows: 8 row loads in flight per lane are enough to cover the latency, and the unrolled body
        // stays within 4 waves per SIMD (all 16 rows hoisted needed 191 VGPRs)
#pragma unroll 1
        for (int g = 0; g < P1_RG; g += 8) {
            float2 f0[8], f1[8];
#pragma unroll
            for (int r = 0; r < 8; r++) ffl_load_pair(flow, w, h, x, y0 + g + r, f0[r], f1[r]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int y = y0 + g + r;
                const int yc = min(y, h - 1);  // rows past the image contribute nothing (predicated below)
                const double dy = (double)y - cy;
                const double wy = pov_mode ? 1.0 : (((double)y > cy) ? wytab[yc] : wytab[h + yc]);
                const double t0 = ((double)f0[r].x * dx0 + (double)f0[r].y * dy) * wx0 * wy;
                const double t1 = ((double)f1[r].x * dx1 + (double)f1[r].y * dy) * wx1 * wy;
                sum += (ok0 && y < h) ? t0 : 0.0;
                sum += (ok1 && y < h) ? t1 : 0.0;
            }
        }
    }
    sum = ffl_wave_sum_f64(sum);
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

void ffl_launch_radial(const RadialTab *rt, int nB, int w, int h, int pov_mode, const double *wytab, double *psum,
                       double *out, hipStream_t st) {
    int nblk = (ffl_strip_waves(w, h, P2_STRIP) + 3) / 4;
    hipLaunchKernelGGL(k_radial, dim3(nblk, nB), dim3(P1_THREADS), 0, st, rt, w, h, pov_mode, wytab, psum);
    hipLaunchKernelGGL(k_radial_final, dim3(nB), dim3(P1_THREADS), 0, st, w, h, nblk, psum, out);
}
