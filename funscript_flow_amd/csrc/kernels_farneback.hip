// gfx950 kernels for dense Farneback flow (what cv2.calcOpticalFlowFarneback computes at
// FunscriptFlow.pyw:878-879).  Built with -ffp-contract=off: every float/double operation below is
// meant literally, in the order written, so that results are bit-identical to the CPU oracle.
//
// Kernel            bound      algorithmic bytes / level pixel
//   k_gray          HBM        4 per full-res pixel (3 in, 1 out)
//   k_pyr_*         HBM / LDS  1 per full-res pixel in + 4 per level pixel out ("k_pyr_level" class): fused 3-tap
//                              kernels for the exact 1x / 2x levels (4 pixels per lane, inside k_pyr_multi), one
//                              LDS-staged pass for the x1/4 and x1/8 levels together (k_pyr_coarse), H + V kernel
//                              pairs for geometries those do not cover
//   k_polyexp       VALU(f64)  24  (4 in, 20 out), 11x11 separable through LDS, f64 accumulators, 4 pixels per
//                              lane; all levels in one merged launch (k_polyexp_multi)
//   k_update_mat    HBM        68  (R0 20 + R1 gather 20 + flow 8 -> M 20); also forms the level's
//                              initial flow (x2 upsample of the coarser level, used from registers)
//   k_blur_solve    HBM        28  (M 20 -> flow 8) [+68 when the next UpdateMatrices is fused; on large
//                              levels the first iteration also runs the level's flow init + UpdateMatrices_0];
//                              the flow is only stored by a level's last iteration (nothing reads it earlier)
// Every UpdateMatrices (standalone, fused update, phase U of the folded first iteration) works one pixel per lane
// with the next row's / item's loads requested before the current one's arithmetic, and sizes its loads by what a
// wave64 load costs the CU's vector-memory path (4 / 8 / 16 bytes per lane: 6.3 / 19.5 / 16.7 cycles): dwords for the
// gathers, 16 bytes for streams, never 8.  DESIGN.md section 4, "Round 2".
#include "ffl_kernels.h"

// ------------------------------------------------------------------------------------------------
// K0: BGR -> gray, OpenCV 8-bit fixed point (15-bit coefficients).  4 pixels per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, int n) {
    int i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(bgr + (size_t)i4 * 3);
        uint32_t a = p[0], b = p[1], c = p[2];
        uint32_t px[12] = {a & 255, (a >> 8) & 255, (a >> 16) & 255, a >> 24, b & 255, (b >> 8) & 255,
                           (b >> 16) & 255, b >> 24, c & 255, (c >> 8) & 255, (c >> 16) & 255, c >> 24};
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t g = (px[j * 3] * 3735u + px[j * 3 + 1] * 19235u + px[j * 3 + 2] * 9798u + 16384u) >> 15;
            o |= g << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(gray + i4) = o;
    } else {
        for (int i = i4; i < n; i++) {
            uint32_t b = bgr[(size_t)i * 3], g = bgr[(size_t)i * 3 + 1], r = bgr[(size_t)i * 3 + 2];
            gray[i] = (uint8_t)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15);
        }
    }
}

void ffl_launch_gray(const uint8_t *bgr, uint8_t *gray, int n_pixels, hipStream_t st) {
    int blocks = (n_pixels + 1023) / 1024;
    hipLaunchKernelGGL(k_gray, dim3(blocks), dim3(256), 0, st, bgr, gray, n_pixels);
}

// ------------------------------------------------------------------------------------------------
// K1: level image  I_k = resize(GaussianBlur(float(gray), ksize, sigma), (lw, lh), INTER_LINEAR)
// The blur is evaluated only at the (up to 4) full-resolution pixels each output samples:
// horizontal pass first (float, symmetric form, REFLECT_101), then vertical, then the two lerps.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ffl_reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// `scale` = (double)src / dst, formed once on the host (IEEE division: the same double the oracle forms)
__device__ __forceinline__ void ffl_resize_coord(int d, int src, double scale, int &i0, int &i1, float &f) {
    // (the x2 upsample of an even-sized level takes ffl_resize_coord_half below; useless while the phase waited for its
    // loads, -1.5 % once it did not)
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
    i0 = sx;
    i1 = sx + 1 < src ? sx + 1 : src - 1;
    f = fx;
}

// the same for scale == 0.5 exactly (a x2 upsample of an even-sized level): (d + 0.5) * 0.5 - 0.5 = d/2 - 0.25 is exact in
// double and in float, so floor and fraction follow from the parity of d -- identical results without the f64 arithmetic
__device__ __forceinline__ void ffl_resize_coord_half(int d, int src, int &i0, int &i1, float &f) {
    int sx = (d - 1) >> 1;
    float fx = (d & 1) ? 0.25f : 0.75f;
    if (d == 0) { sx = 0; fx = 0.f; }
    if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
    i0 = sx;
    i1 = sx + 1 < src ? sx + 1 : src - 1;
    f = fx;
}

// Two plain streaming kernels per level, no LDS and no barriers (the earlier LDS-tiled version spent
// its time in four dependent phases per workgroup and never filled the device at the coarse levels):
//   k_pyr_h  one lane per (row y, output column d, lerp side q): horizontal blur of the full-resolution
//            row at the sampled column -> tmp[y][d][q]            (float, symmetric form, REFLECT_101)
//   k_pyr_v  one lane per output pixel: vertical blur of tmp at the (<= 2) sampled rows for the
//            (<= 2) sampled columns, then the two lerps -> I
// Samples whose lerp weight is exactly 0 are not evaluated: v*1 + s*0 == v for finite s >= 0.
// R > 0: blur radius known at compile time (taps unrolled, coefficients in scalar registers);
// R == 0: runtime radius.  nq = 2 when the level resamples in x (lw != w), else 1.
// rows per lane of k_pyr_h: enough loads in flight to cover the memory latency without spilling
// (a row costs 2R+1 byte loads)
constexpr int ffl_pyr_rows(int R) { return R == 1 ? 16 : (R == 4 ? 8 : 4); }
constexpr int ffl_pyr_vrows(int R) { return R == 1 ? 4 : 1; }  // output rows per lane of k_pyr_v
template <int R>
__global__ __launch_bounds__(256) void k_pyr_h(const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut, int w,
                                               int h, int lw, int nq, double sx, GaussKernel gk,
                                               float *__restrict__ tmp, size_t tmp_stride) {
    // each lane produces ROWS rows of its column (one workgroup per 256 outputs would be bound by the
    // workgroup dispatch rate, not by memory)
    constexpr int ROWS = ffl_pyr_rows(R);
    const int i = blockIdx.x * 256 + threadIdx.x, u = blockIdx.z;
    if (i >= lw * nq) return;
    const int d = nq == 2 ? i >> 1 : i, q = nq == 2 ? (i & 1) : 0;
    const int r = R > 0 ? R : (gk.ksize >> 1);
    int x0, x1;
    float fx;
    ffl_resize_coord(d, w, sx, x0, x1, fx);
    const bool on = q == 0 || fx != 0.f;
    const int cx = q ? x1 : x0;
    const bool interior = cx >= r && cx + r < w;
    int xm[R > 0 ? R : 1], xp[R > 0 ? R : 1];  // reflected tap columns (border lanes, compile-time radius)
    if (R > 0) {
#pragma unroll
        for (int t = 1; t <= R; t++) {
            xm[t - 1] = ffl_reflect101(cx - t, w);
            xp[t - 1] = ffl_reflect101(cx + t, w);
        }
    }
    const uint8_t *img = gray_base + (size_t)ut->fslot[u] * gray_stride;
    float *out = tmp + (size_t)u * tmp_stride + i;
    const int ybase = blockIdx.y * ROWS;
    // word-wise tap fetch needs 4-byte aligned rows (w % 4 == 0; frame slots are w*h apart) and all
    // taps inside the row; the last word may reach <= 3 bytes past the taps, still inside the slot array
    // (the gray buffer is allocated with 16 bytes of slack)
    const int wfirst = cx - r, woff = wfirst & 3, wbase = wfirst - woff;
    const bool wide = R > 0 && interior && (w & 3) == 0;
    // fixed trip count, rows clamped for the loads and predicated for the store: the loop unrolls and
    // the loads of all rows are in flight together (a rolled loop pays one memory latency per row)
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int y = min(ybase + k, h - 1);
        float acc = 0.f;
        if (on) {
            const uint8_t *row = img + (size_t)y * w;
            if (R > 0 && wide) {
                // interior lane: the 2R+1 taps sit in NW aligned 32-bit words -> NW loads instead of
                // 2R+1 byte loads (the texture-address unit is paid per wave-instruction, not per byte),
                // re-aligned with v_alignbyte, bytes converted with v_cvt_f32_ubyteN
                constexpr int NW = (3 + 2 * R + 1 + 3) / 4;
                const uint32_t *wp = reinterpret_cast<const uint32_t *>(row + wbase);
                uint32_t wd[NW], a[NW];
#pragma unroll
                for (int q2 = 0; q2 < NW; q2++) wd[q2] = wp[q2];
#pragma unroll
                for (int q2 = 0; q2 < NW; q2++)
                    a[q2] = q2 + 1 < NW ? __builtin_amdgcn_alignbyte(wd[q2 + 1], wd[q2], woff) : wd[q2] >> (8 * woff);
                auto tap = [&](int j) { return (float)((a[j >> 2] >> (8 * (j & 3))) & 255u); };
                acc = gk.k[R] * tap(R);
#pragma unroll
                for (int t = 1; t <= R; t++) acc = acc + gk.k[R + t] * (tap(R - t) + tap(R + t));
            } else if (R > 0) {
                acc = gk.k[r] * (float)row[cx];
#pragma unroll
                for (int t = 1; t <= R; t++) acc = acc + gk.k[R + t] * ((float)row[xm[t - 1]] + (float)row[xp[t - 1]]);
            } else if (interior) {
                acc = gk.k[r] * (float)row[cx];
                for (int t = 1; t <= r; t++) acc = acc + gk.k[r + t] * ((float)row[cx - t] + (float)row[cx + t]);
            } else {
                acc = gk.k[r] * (float)row[cx];
                for (int t = 1; t <= r; t++)
                    acc = acc + gk.k[r + t] * ((float)row[ffl_reflect101(cx - t, w)] + (float)row[ffl_reflect101(cx + t, w)]);
            }
        }
        if (ybase + k < h) out[(size_t)y * lw * nq] = acc;
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_pyr_v(const float *__restrict__ tmp, size_t tmp_stride, int w, int h, int lw,
                                               int lh, int nq, double sx, double sy, GaussKernel gk,
                                               float *__restrict__ I, size_t I_stride) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), u = blockIdx.z;
    if (dx >= lw) return;
    const int r = R > 0 ? R : (gk.ksize >> 1);
    int x0, x1;
    float a1;
    ffl_resize_coord(dx, w, sx, x0, x1, a1);
    const float a0 = 1.f - a1;
    const size_t pitch = (size_t)lw * nq;
    const float *col = tmp + (size_t)u * tmp_stride + (size_t)dx * nq;  // [row * pitch + q]
    // 64 x (4 * ROWS) outputs per workgroup; fixed trip count + clamped rows so that the loop unrolls
    // and every row's loads are in flight together
    constexpr int ROWS = ffl_pyr_vrows(R);
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
    const int dyy = blockIdx.y * (4 * ROWS) + (threadIdx.x >> 6) + 4 * k;
    const int dy = min(dyy, lh - 1);
    int y0, y1;
    float b1;
    ffl_resize_coord(dy, h, sy, y0, y1, b1);
    const float b0 = 1.f - b1;
    float t[2] = {0.f, 0.f};
#pragma unroll
    for (int qy = 0; qy < 2; qy++) {
        if (qy == 1 && b1 == 0.f) break;
        const int cy = qy ? y1 : y0;
        const bool interior = cy >= r && cy + r < h;
        float v[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q == 1 && a1 == 0.f) break;
            const float *p = col + q;
            float acc = gk.k[r] * p[(size_t)cy * pitch];
            if (interior) {
#pragma unroll
                for (int j = 1; j <= r; j++)
                    acc = acc + gk.k[r + j] * (p[(size_t)(cy - j) * pitch] + p[(size_t)(cy + j) * pitch]);
            } else {
#pragma unroll
                for (int j = 1; j <= r; j++)
                    acc = acc + gk.k[r + j] * (p[(size_t)ffl_reflect101(cy - j, h) * pitch] +
                                               p[(size_t)ffl_reflect101(cy + j, h) * pitch]);
            }
            v[q] = acc;
        }
        t[qy] = v[0] * a0 + v[1] * a1;
    }
    if (dyy < lh) I[(size_t)u * I_stride + (size_t)dy * lw + dx] = t[0] * b0 + t[1] * b1;
    }
}

// Resampling levels (nq == 2), compile-time radius: one lane computes BOTH lerp sides of an output
// column in k_pyr_h2 (their 2R+2 source bytes overlap: one set of word loads, one 8-byte store), and
// k_pyr_v2 fetches the 2R+2 distinct rows its two vertical sums share once, 8 bytes (both sides) per
// load -- 10 / 20 loads per output pixel at R = 4 / 9 instead of 36 / 76.  Same operations in the same
// order as k_pyr_h / k_pyr_v; lanes at the image border fall back to the per-sample form.
template <int R>
__device__ __forceinline__ void k_pyr_h2_body(const unsigned bx, const unsigned by, const unsigned bz, const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut, int w,
                                                int h, int lw, double sx, GaussKernel gk, float *__restrict__ tmp,
                                                size_t tmp_stride) {
    constexpr int ROWS = ffl_pyr_rows(R);
    const int d = bx * 256 + threadIdx.x, u = bz;
    if (d >= lw) return;
    int x0, x1;
    float fx;
    ffl_resize_coord(d, w, sx, x0, x1, fx);
    const bool on1 = fx != 0.f;
    // both samples from one run of aligned words: taps x0-R .. x0+R+1 inside the row, x1 == x0 + 1
    const bool wide = x1 == x0 + 1 && x0 >= R && x1 + R < w && (w & 3) == 0;
    const int wfirst = x0 - R, woff = wfirst & 3, wbase = wfirst - woff;
    int xm[2][R], xp[2][R];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int t = 1; t <= R; t++) {
            xm[q][t - 1] = ffl_reflect101((q ? x1 : x0) - t, w);
            xp[q][t - 1] = ffl_reflect101((q ? x1 : x0) + t, w);
        }
    const uint8_t *img = gray_base + (size_t)ut->fslot[u] * gray_stride;
    float2 *out = reinterpret_cast<float2 *>(tmp + (size_t)u * tmp_stride) + d;
    const int ybase = by * ROWS;
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int y = min(ybase + k, h - 1);
        const uint8_t *row = img + (size_t)y * w;
        float acc0, acc1 = 0.f;
        if (wide) {
            constexpr int NW = (3 + 2 * R + 2 + 3) / 4;
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(row + wbase);
            uint32_t wd[NW], a[NW];
#pragma unroll
            for (int q2 = 0; q2 < NW; q2++) wd[q2] = wp[q2];
#pragma unroll
            for (int q2 = 0; q2 < NW; q2++)
                a[q2] = q2 + 1 < NW ? __builtin_amdgcn_alignbyte(wd[q2 + 1], wd[q2], woff) : wd[q2] >> (8 * woff);
            auto tap = [&](int j) { return (float)((a[j >> 2] >> (8 * (j & 3))) & 255u); };
            acc0 = gk.k[R] * tap(R);
#pragma unroll
            for (int t = 1; t <= R; t++) acc0 = acc0 + gk.k[R + t] * (tap(R - t) + tap(R + t));
            acc1 = gk.k[R] * tap(R + 1);
#pragma unroll
            for (int t = 1; t <= R; t++) acc1 = acc1 + gk.k[R + t] * (tap(R + 1 - t) + tap(R + 1 + t));
        } else {
            acc0 = gk.k[R] * (float)row[x0];
#pragma unroll
            for (int t = 1; t <= R; t++) acc0 = acc0 + gk.k[R + t] * ((float)row[xm[0][t - 1]] + (float)row[xp[0][t - 1]]);
            if (on1) {
                acc1 = gk.k[R] * (float)row[x1];
#pragma unroll
                for (int t = 1; t <= R; t++)
                    acc1 = acc1 + gk.k[R + t] * ((float)row[xm[1][t - 1]] + (float)row[xp[1][t - 1]]);
            }
        }
        if (ybase + k < h) out[(size_t)y * lw] = make_float2(acc0, on1 ? acc1 : 0.f);
    }
}
template <int R>
__global__ __launch_bounds__(256) void k_pyr_h2(const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut, int w,
                                                int h, int lw, double sx, GaussKernel gk, float *__restrict__ tmp,
                                                size_t tmp_stride) {
    k_pyr_h2_body<R>(blockIdx.x, blockIdx.y, blockIdx.z, gray_base, gray_stride, ut, w, h, lw, sx, gk, tmp, tmp_stride);
}

template <int R>
__device__ __forceinline__ void k_pyr_v2_body(const unsigned bx, const unsigned by, const unsigned bz, const float *__restrict__ tmp, size_t tmp_stride, int w, int h, int lw,
                                                int lh, double sx, double sy, GaussKernel gk, float *__restrict__ I,
                                                size_t I_stride) {
    const int dx = bx * 64 + (threadIdx.x & 63), u = bz;
    if (dx >= lw) return;
    int x0, x1;
    float a1;
    ffl_resize_coord(dx, w, sx, x0, x1, a1);
    const float a0 = 1.f - a1;
    const float2 *col = reinterpret_cast<const float2 *>(tmp + (size_t)u * tmp_stride) + dx;  // [row * lw]
    const int dyy = by * 4 + (threadIdx.x >> 6);
    const int dy = min(dyy, lh - 1);
    int y0, y1;
    float b1;
    ffl_resize_coord(dy, h, sy, y0, y1, b1);
    const float b0 = 1.f - b1;
    float t[2] = {0.f, 0.f};
    if (y1 == y0 + 1 && y0 >= R && y1 + R < h) {
        float2 V[2 * R + 2];  // rows y0-R .. y1+R, both lerp sides
#pragma unroll
        for (int j = 0; j < 2 * R + 2; j++) V[j] = col[(size_t)(y0 - R + j) * lw];
#pragma unroll
        for (int qy = 0; qy < 2; qy++) {
            if (qy == 1 && b1 == 0.f) break;
            const int c = R + qy;
            float v0 = gk.k[R] * V[c].x, v1 = gk.k[R] * V[c].y;
#pragma unroll
            for (int j = 1; j <= R; j++) {
                v0 = v0 + gk.k[R + j] * (V[c - j].x + V[c + j].x);
                v1 = v1 + gk.k[R + j] * (V[c - j].y + V[c + j].y);
            }
            t[qy] = v0 * a0 + (a1 == 0.f ? 0.f : v1) * a1;
        }
    } else {
#pragma unroll
        for (int qy = 0; qy < 2; qy++) {
            if (qy == 1 && b1 == 0.f) break;
            const int cy = qy ? y1 : y0;
            float2 c = col[(size_t)cy * lw];
            float v0 = gk.k[R] * c.x, v1 = gk.k[R] * c.y;
#pragma unroll
            for (int j = 1; j <= R; j++) {
                const float2 m = col[(size_t)ffl_reflect101(cy - j, h) * lw], p = col[(size_t)ffl_reflect101(cy + j, h) * lw];
                v0 = v0 + gk.k[R + j] * (m.x + p.x);
                v1 = v1 + gk.k[R + j] * (m.y + p.y);
            }
            t[qy] = v0 * a0 + (a1 == 0.f ? 0.f : v1) * a1;
        }
    }
    if (dyy < lh) I[(size_t)u * I_stride + (size_t)dy * lw + dx] = t[0] * b0 + t[1] * b1;
}
template <int R>
__global__ __launch_bounds__(256) void k_pyr_v2(const float *__restrict__ tmp, size_t tmp_stride, int w, int h, int lw,
                                                int lh, double sx, double sy, GaussKernel gk, float *__restrict__ I,
                                                size_t I_stride) {
    k_pyr_v2_body<R>(blockIdx.x, blockIdx.y, blockIdx.z, tmp, tmp_stride, w, h, lw, lh, sx, sy, gk, I, I_stride);
}

// Fused form for the two fine levels of every BASELINE size: 3-tap blur (R = 1) with an exact S = 1
// (level 0) or S = 2 (level 1) decimation.  One lane produces FROWS consecutive output rows of one
// output column: the S*FROWS + 2 source rows are fetched word-wise once, blurred horizontally in
// registers, then combined vertically -- no intermediate plane, same operations in the same order as
// k_pyr_h + k_pyr_v (for S = 2 the lerp weights are exactly 0.5, for S = 1 the lerps are identities).
#define FFL_PYR_FROWS 4
#ifndef FFL_FR1
#define FFL_FR1 8  // output rows per lane of the fused 3-tap kernels, level 0 / level 1 (33 x 1080p frames: 4/2 136 us,
#define FFL_FR2 2  // 8/2 129, 12/2 136, 16/2 208, 8/4 256, 8/1 157, 2/2 175)
#endif
template <int S>
__global__ __launch_bounds__(256) void k_pyr_fused3(const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut,
                                                    int w, int h, int lw, int lh, GaussKernel gk,
                                                    float *__restrict__ I, size_t I_stride) {
    constexpr int NR = S * FFL_PYR_FROWS + 2;  // source rows per lane
    const int dx = blockIdx.x * 256 + threadIdx.x, u = blockIdx.z;
    if (dx >= lw) return;
    const uint8_t *img = gray_base + (size_t)ut->fslot[u] * gray_stride;
    const int cx = S * dx;                       // first sampled column; S == 2 also samples cx + 1
    const int first = cx - 1, woff = first & 3, wbase = first - woff;
    const bool wide = first >= 0 && cx + S < w && (w & 3) == 0;  // taps cx-1 .. cx+S inside the row
    const float k0 = gk.k[1], k1 = gk.k[2];
    const int dy0 = blockIdx.y * FFL_PYR_FROWS;
    float H0[NR], H1[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int sy = ffl_reflect101(min(S * dy0 - 1 + j, h + 1), h);  // rows past the image are never used
        const uint8_t *row = img + (size_t)sy * w;
        float b[S + 2];
        if (wide) {
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(row + wbase);
            const uint32_t lo = wp[0], hi = wp[1];
            const uint32_t a = __builtin_amdgcn_alignbyte(hi, lo, woff);  // bytes first .. first+3
#pragma unroll
            for (int t = 0; t < S + 2; t++) b[t] = (float)((a >> (8 * t)) & 255u);
        } else {
#pragma unroll
            for (int t = 0; t < S + 2; t++) b[t] = (float)row[ffl_reflect101(first + t, w)];
        }
        H0[j] = k0 * b[1] + k1 * (b[0] + b[2]);
        H1[j] = S == 2 ? k0 * b[2] + k1 * (b[1] + b[3]) : 0.f;
    }
#pragma unroll
    for (int o = 0; o < FFL_PYR_FROWS; o++) {
        const int dy = dy0 + o;
        if (dy >= lh) break;
        const int c = S * o + 1;  // local index of source row S*dy
        float out;
        if (S == 1) {
            const float v00 = k0 * H0[c] + k1 * (H0[c - 1] + H0[c + 1]);
            out = (v00 * 1.f + 0.f * 0.f) * 1.f + 0.f * 0.f;
        } else {
            const float v00 = k0 * H0[c] + k1 * (H0[c - 1] + H0[c + 1]);
            const float v01 = k0 * H1[c] + k1 * (H1[c - 1] + H1[c + 1]);
            const float v10 = k0 * H0[c + 1] + k1 * (H0[c] + H0[c + 2]);
            const float v11 = k0 * H1[c + 1] + k1 * (H1[c] + H1[c + 2]);
            const float t0 = v00 * 0.5f + v01 * 0.5f, t1 = v10 * 0.5f + v11 * 0.5f;
            out = t0 * 0.5f + t1 * 0.5f;
        }
        I[(size_t)u * I_stride + (size_t)dy * lw + dx] = out;
    }
}

// The same, four adjacent output pixels per lane (w and lw multiples of 4): the 4S + 2 source bytes of a
// row come with ONE 12- / 16-byte load and the four results leave with one 16-byte store -- a sixth of
// the vector-memory instructions of the pixel-per-lane kernel above, which spent its time issuing them.
template <int S, int FR>
__device__ __forceinline__ void k_pyr_fused3x4_body(const unsigned bx, const unsigned by, const unsigned bz, const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut,
                                                      int w, int h, int lw, int lh, GaussKernel gk,
                                                      float *__restrict__ I, size_t I_stride) {
    constexpr int NR = S * FR + 2;   // source rows per lane
    constexpr int NBY = 4 * S + 2;   // source bytes per row: columns cx-1 .. cx+4S
    const int dx = 4 * (bx * 256 + threadIdx.x), u = bz;
    if (dx >= lw) return;
    const uint8_t *img = gray_base + (size_t)ut->fslot[u] * gray_stride;
    const int cx = S * dx;  // first sampled column, a multiple of 4
    const bool wide = cx >= 4 && cx + 4 * S + 4 <= w;  // the aligned words cx-4 .. cx+4S+3 are inside the row
    const float k0 = gk.k[1], k1 = gk.k[2];
    const int dy0 = by * FR;
    float H[NR][4 * S];  // horizontal blur at columns cx .. cx+4S-1
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int sy = ffl_reflect101(min(S * dy0 - 1 + j, h + 1), h);  // rows past the image are never used
        const uint8_t *row = img + (size_t)sy * w;
        float b[NBY];
        if (wide) {
            uint32_t wd[S + 2];
            if (S == 1) {
                struct __attribute__((packed, aligned(4))) u3 { uint32_t a, b, c; };
                const u3 t = *reinterpret_cast<const u3 *>(row + cx - 4);
                wd[0] = t.a; wd[1] = t.b; wd[2] = t.c;
            } else {
                struct __attribute__((packed, aligned(4))) u4 { uint32_t a, b, c, d; };
                const u4 t = *reinterpret_cast<const u4 *>(row + cx - 4);
                wd[0] = t.a; wd[1] = t.b; wd[2] = t.c; wd[3] = t.d;
            }
            b[0] = (float)(wd[0] >> 24);  // column cx-1
#pragma unroll
            for (int t = 0; t < 4 * S; t++) b[1 + t] = (float)((wd[1 + (t >> 2)] >> (8 * (t & 3))) & 255u);
            b[NBY - 1] = (float)(wd[S + 1] & 255u);  // column cx+4S
        } else {
#pragma unroll
            for (int t = 0; t < NBY; t++) b[t] = (float)row[ffl_reflect101(cx - 1 + t, w)];
        }
#pragma unroll
        for (int c = 0; c < 4 * S; c++) H[j][c] = k0 * b[c + 1] + k1 * (b[c] + b[c + 2]);
    }
#pragma unroll
    for (int o = 0; o < FR; o++) {
        const int dy = dy0 + o;
        if (dy >= lh) break;
        const int c = S * o + 1;  // local index of source row S*dy
        float out[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (S == 1) {
                const float v00 = k0 * H[c][i] + k1 * (H[c - 1][i] + H[c + 1][i]);
                out[i] = (v00 * 1.f + 0.f * 0.f) * 1.f + 0.f * 0.f;
            } else {
                const float v00 = k0 * H[c][2 * i] + k1 * (H[c - 1][2 * i] + H[c + 1][2 * i]);
                const float v01 = k0 * H[c][2 * i + 1] + k1 * (H[c - 1][2 * i + 1] + H[c + 1][2 * i + 1]);
                const float v10 = k0 * H[c + 1][2 * i] + k1 * (H[c][2 * i] + H[c + 2][2 * i]);
                const float v11 = k0 * H[c + 1][2 * i + 1] + k1 * (H[c][2 * i + 1] + H[c + 2][2 * i + 1]);
                const float t0 = v00 * 0.5f + v01 * 0.5f, t1 = v10 * 0.5f + v11 * 0.5f;
                out[i] = t0 * 0.5f + t1 * 0.5f;
            }
        }
        ffl_f4u t;
        t.x = out[0]; t.y = out[1]; t.z = out[2]; t.w = out[3];
        *reinterpret_cast<ffl_f4u *>(I + (size_t)u * I_stride + (size_t)dy * lw + dx) = t;
    }
}
template <int S, int FR>
__global__ __launch_bounds__(256) void k_pyr_fused3x4(const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut,
                                                      int w, int h, int lw, int lh, GaussKernel gk,
                                                      float *__restrict__ I, size_t I_stride) {
    k_pyr_fused3x4_body<S, FR>(blockIdx.x, blockIdx.y, blockIdx.z, gray_base, gray_stride, ut, w, h, lw, lh, gk, I, I_stride);
}

// ------------------------------------------------------------------------------------------------
// The two coarse levels (x1/4 with a 9-tap blur, x1/8 with a 19-tap blur) from ONE pass over the gray frame:
// a workgroup stages a 128 x 64 source tile (+ the 19-tap halo, REFLECT_101 applied while loading) in LDS with
// coalesced dword loads, forms the horizontal blur at the sampled columns of BOTH levels into LDS, then the
// vertical blur at the sampled rows and the two lerps.  No intermediate plane in memory (the H + V kernel pairs
// above wrote and re-read 6 MB per 1080p frame) and one read of the frame instead of two strided ones.
// Exact x4 / x8 decimation only (w, h multiples of 8): output d samples the source columns S*d + S/2 - 1 and
// S*d + S/2 with weights 0.5 / 0.5 -- what ffl_resize_coord yields for these scales -- so the operations and
// their order are those of k_pyr_h2 / k_pyr_v2.
// ------------------------------------------------------------------------------------------------
#define PC_TW 128
#define PC_TH 64
#define PC_PW 36   // staged tile: dwords per row (columns X0-8 .. X0+135)
#define PC_PH 76   // rows Y0-6 .. Y0+69
#define PC_H2R 70  // level-2 H rows: Y0-3 .. Y0+66
#define PC_H2P 66  // pitch of sH2 (64 sampled columns + pad)
#define PC_H3P 34  // pitch of sH3 (32 sampled columns + pad)
__global__ __launch_bounds__(256) void k_pyr_coarse(const uint8_t *__restrict__ gray_base, size_t gray_stride,
                                                    const UTab *__restrict__ ut, int w, int h, GaussKernel gk2,
                                                    GaussKernel gk3, float *__restrict__ I2, size_t I2_stride,
                                                    float *__restrict__ I3, size_t I3_stride, int tiles_x, int tiles_y,
                                                    int nU) {
    __shared__ __attribute__((aligned(8))) uint32_t sP[PC_PH][PC_PW];
    __shared__ __attribute__((aligned(8))) float sH2[PC_H2R][PC_H2P];
    __shared__ __attribute__((aligned(8))) float sH3[PC_PH][PC_H3P];
    const int tid = threadIdx.x;
    const unsigned per = (unsigned)tiles_x * tiles_y;
    unsigned tt;  // XCD-aware order: neighbouring tiles (which share the 19-tap halo) run on one XCD
    if (!ffl_xcd_tile(blockIdx.x, per * nU, tt)) return;
    const int u = tt / per, t = tt - u * per, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int X0 = tx * PC_TW, Y0 = ty * PC_TH;
    const uint8_t *img = gray_base + (size_t)ut->fslot[u] * gray_stride;
    // ---- stage the tile: 16 bytes per lane and load (the row segment X0-8 .. X0+135 is 9 such pieces), fixed trip
    // count so that all loads of a lane are in flight together
    constexpr int PCQ = PC_PW / 4, NLOAD = (PC_PH * PCQ + 255) / 256;
    struct __attribute__((packed, aligned(4))) u4 { uint32_t a, b, c, d; };
#pragma unroll
    for (int k = 0; k < NLOAD; k++) {
        const int i = tid + 256 * k;
        if (i < PC_PH * PCQ) {
            const int ry = i / PCQ, q = i - ry * PCQ;
            const int gy = ffl_reflect101(Y0 - 6 + ry, h), X = X0 - 8 + 16 * q;
            const uint8_t *row = img + (size_t)gy * w;
            uint32_t v[4];
            if (X >= 0 && X + 15 < w) {
                const u4 t4 = *reinterpret_cast<const u4 *>(row + X);
                v[0] = t4.a; v[1] = t4.b; v[2] = t4.c; v[3] = t4.d;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int Xc = X + 4 * c;
                    if (Xc >= 0 && Xc + 3 < w)  // w % 4 == 0: a dword is either inside the row or entirely outside
                        v[c] = *reinterpret_cast<const uint32_t *>(row + Xc);
                    else
                        v[c] = (uint32_t)row[ffl_reflect101(Xc, w)] | ((uint32_t)row[ffl_reflect101(Xc + 1, w)] << 8) |
                               ((uint32_t)row[ffl_reflect101(Xc + 2, w)] << 16) | ((uint32_t)row[ffl_reflect101(Xc + 3, w)] << 24);
                }
            }
            *reinterpret_cast<uint2 *>(&sP[ry][4 * q]) = make_uint2(v[0], v[1]);
            *reinterpret_cast<uint2 *>(&sP[ry][4 * q + 2]) = make_uint2(v[2], v[3]);
        }
    }
    __syncthreads();
    // ---- horizontal pass, level 2 (radius 4): output column dl samples tile bytes 4*dl+9, 4*dl+10 (image columns
    // 4d+1, 4d+2); taps of both = bytes 4*dl+5 .. 4*dl+14 = dwords dl+1 .. dl+3
    // (fixed trip counts + a row predicate: the loops unroll and a lane's LDS reads are in flight together)
    {
        const int dl = tid & 31, rr = tid >> 5;
#pragma unroll 3
        for (int k = 0; k < (PC_H2R + 7) / 8; k++) {
            const int ry = rr + 8 * k;
            if (ry < PC_H2R) {
                const uint32_t *p = &sP[ry + 3][dl + 1];
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                auto tap = [&](int j) {  // byte j of the 12 fetched ones; the first sample's centre is byte 5
                    const uint32_t wd = j < 4 ? d0 : (j < 8 ? d1 : d2);
                    return (float)((wd >> (8 * (j & 3))) & 255u);
                };
                float a0 = gk2.k[4] * tap(5), a1 = gk2.k[4] * tap(6);
#pragma unroll
                for (int j = 1; j <= 4; j++) {
                    a0 = a0 + gk2.k[4 + j] * (tap(5 - j) + tap(5 + j));
                    a1 = a1 + gk2.k[4 + j] * (tap(6 - j) + tap(6 + j));
                }
                *reinterpret_cast<float2 *>(&sH2[ry][2 * dl]) = make_float2(a0, a1);
            }
        }
    }
    // ---- horizontal pass, level 3 (radius 9): output column dl samples tile bytes 8*dl+11, 8*dl+12 (image columns
    // 8d+3, 8d+4); taps of both = bytes 8*dl+2 .. 8*dl+21 = dwords 2*dl .. 2*dl+5
    {
        const int dl = tid & 15, rr = tid >> 4;
#pragma unroll
        for (int k = 0; k < (PC_PH + 15) / 16; k++) {
            const int ry = rr + 16 * k;
            if (ry < PC_PH) {
                const uint32_t *p = &sP[ry][2 * dl];
                uint32_t d[6];
#pragma unroll
                for (int q = 0; q < 6; q++) d[q] = p[q];
                auto tap = [&](int j) { return (float)((d[j >> 2] >> (8 * (j & 3))) & 255u); };  // byte j of the 24
                float a0 = gk3.k[9] * tap(11), a1 = gk3.k[9] * tap(12);
#pragma unroll
                for (int j = 1; j <= 9; j++) {
                    a0 = a0 + gk3.k[9 + j] * (tap(11 - j) + tap(11 + j));
                    a1 = a1 + gk3.k[9 + j] * (tap(12 - j) + tap(12 + j));
                }
                *reinterpret_cast<float2 *>(&sH3[ry][2 * dl]) = make_float2(a0, a1);
            }
        }
    }
    __syncthreads();
    // ---- vertical pass + lerps, level 2: 32 x 16 outputs, two per lane.  Output row el samples image rows
    // 4e+1, 4e+2 = sH2 rows 4*el+4, 4*el+5; lerp weights are exactly 0.5 (a0 = 1 - a1 as in ffl_resize_coord's users)
    {
        const float a1 = 0.5f, a0 = 1.f - a1, b1 = 0.5f, b0 = 1.f - b1;
        const int lw2 = w >> 2, lh2 = h >> 2;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int o = tid + 256 * k, el = o >> 5, dl = o & 31;
            const int dx = (X0 >> 2) + dl, dy = (Y0 >> 2) + el;
            float2 V[10];  // sH2 rows 4*el .. 4*el+9
#pragma unroll
            for (int j = 0; j < 10; j++) V[j] = *reinterpret_cast<const float2 *>(&sH2[4 * el + j][2 * dl]);
            float tq[2];
#pragma unroll
            for (int qy = 0; qy < 2; qy++) {
                const int c = 4 + qy;
                float v0 = gk2.k[4] * V[c].x, v1 = gk2.k[4] * V[c].y;
#pragma unroll
                for (int j = 1; j <= 4; j++) {
                    v0 = v0 + gk2.k[4 + j] * (V[c - j].x + V[c + j].x);
                    v1 = v1 + gk2.k[4 + j] * (V[c - j].y + V[c + j].y);
                }
                tq[qy] = v0 * a0 + v1 * a1;
            }
            if (dx < lw2 && dy < lh2) I2[(size_t)u * I2_stride + (size_t)dy * lw2 + dx] = tq[0] * b0 + tq[1] * b1;
        }
    }
    // ---- level 3: 16 x 8 outputs on the first 128 lanes.  Output row el samples image rows 8e+3, 8e+4 = sH3 rows
    // 8*el+9, 8*el+10
    if (tid < 128) {
        const float a1 = 0.5f, a0 = 1.f - a1, b1 = 0.5f, b0 = 1.f - b1;
        const int lw3 = w >> 3, lh3 = h >> 3;
        const int el = tid >> 4, dl = tid & 15;
        const int dx = (X0 >> 3) + dl, dy = (Y0 >> 3) + el;
        float2 V[20];  // sH3 rows 8*el .. 8*el+19
#pragma unroll
        for (int j = 0; j < 20; j++) V[j] = *reinterpret_cast<const float2 *>(&sH3[8 * el + j][2 * dl]);
        float tq[2];
#pragma unroll
        for (int qy = 0; qy < 2; qy++) {
            const int c = 9 + qy;
            float v0 = gk3.k[9] * V[c].x, v1 = gk3.k[9] * V[c].y;
#pragma unroll
            for (int j = 1; j <= 9; j++) {
                v0 = v0 + gk3.k[9 + j] * (V[c - j].x + V[c + j].x);
                v1 = v1 + gk3.k[9 + j] * (V[c - j].y + V[c + j].y);
            }
            tq[qy] = v0 * a0 + v1 * a1;
        }
        if (dx < lw3 && dy < lh3) I3[(size_t)u * I3_stride + (size_t)dy * lw3 + dx] = tq[0] * b0 + tq[1] * b1;
    }
}

// true when the x1/4 and x1/8 levels of a w x h frame can take the one-pass kernel
static bool ffl_pyr_coarse_ok(int w, int h, const PyrJob &l2, const PyrJob &l3) {
    return !((w & 7) || (h & 7) || l2.lw != w / 4 || l2.lh != h / 4 || l3.lw != w / 8 || l3.lh != h / 8 ||
             l2.gk.ksize != 9 || l3.gk.ksize != 19);
}
bool ffl_launch_pyr_coarse(const uint8_t *gray_base, size_t gray_stride, const UTab *ut, int nU, int w, int h,
                           const PyrJob &l2, const PyrJob &l3, hipStream_t st) {
    if (!ffl_pyr_coarse_ok(w, h, l2, l3)) return false;
    const int tiles_x = (w + PC_TW - 1) / PC_TW, tiles_y = (h + PC_TH - 1) / PC_TH;
    hipLaunchKernelGGL(k_pyr_coarse, dim3(ffl_xcd_blocks((unsigned)tiles_x * tiles_y * nU)), dim3(256), 0, st, gray_base,
                       gray_stride, ut, w, h, l2.gk, l3.gk, l2.I, l2.I_stride, l3.I, l3.I_stride, tiles_x, tiles_y, nU);
    return true;
}

size_t ffl_pyr_tmp_floats(int w, int h, int lw) { return (size_t)h * lw * (lw != w ? 2 : 1); }

// All levels' pyramid work in TWO launches (1-D grids cut into per-job ranges): phase A = the fused fine
// levels + the horizontal passes of the resampling levels, phase B = their vertical passes.  Six small,
// latency-bound launches (117 us back to back at 1080p) overlap inside two.
__global__ __launch_bounds__(256) void k_pyr_multi(const uint8_t *__restrict__ gray_base, size_t gray_stride, const UTab *__restrict__ ut,
                                                   PyrJobs jobs) {
    int i = 0;
#pragma unroll
    for (int t = 1; t < FFL_MAX_JOBS; t++)
        if (t < jobs.n && blockIdx.x >= jobs.j[t].first) i = t;
    const PyrJob &J = jobs.j[i];
    unsigned t;  // XCD-aware order inside a job: vertically adjacent blocks (which share source rows) run on one XCD
    if (!ffl_xcd_tile(blockIdx.x - J.first, J.count, t)) return;
    const unsigned per = J.gx * J.gy;
    const unsigned bz = t / per, r = t - bz * per, by = r / J.gx, bx = r - by * J.gx;
    switch (J.kind) {
        case FFL_PYR_F1: k_pyr_fused3x4_body<1, FFL_FR1>(bx, by, bz, gray_base, gray_stride, ut, J.w, J.h, J.lw, J.lh, J.gk, J.I, J.I_stride); break;
        case FFL_PYR_F2: k_pyr_fused3x4_body<2, FFL_FR2>(bx, by, bz, gray_base, gray_stride, ut, J.w, J.h, J.lw, J.lh, J.gk, J.I, J.I_stride); break;
        case FFL_PYR_H4: k_pyr_h2_body<4>(bx, by, bz, gray_base, gray_stride, ut, J.w, J.h, J.lw, J.sx, J.gk, J.tmp, J.tmp_stride); break;
        case FFL_PYR_H9: k_pyr_h2_body<9>(bx, by, bz, gray_base, gray_stride, ut, J.w, J.h, J.lw, J.sx, J.gk, J.tmp, J.tmp_stride); break;
        case FFL_PYR_V4: k_pyr_v2_body<4>(bx, by, bz, J.tmp, J.tmp_stride, J.w, J.h, J.lw, J.lh, J.sx, J.sy, J.gk, J.I, J.I_stride); break;
        default: k_pyr_v2_body<9>(bx, by, bz, J.tmp, J.tmp_stride, J.w, J.h, J.lw, J.lh, J.sx, J.sy, J.gk, J.I, J.I_stride); break;
    }
}

// kind of merged job a level maps to in phase A (-1: the level needs the generic per-level kernels)
static int ffl_pyr_kind(int w, int h, int lw, int lh, int ksize) {
    const int r = ksize / 2;
    if ((w & 3) || (lw & 3)) return -1;
    if (r == 1 && lw == w && lh == h) return FFL_PYR_F1;
    if (r == 1 && w == 2 * lw && h == 2 * lh) return FFL_PYR_F2;
    if (lw != w && r == 4) return FFL_PYR_H4;
    if (lw != w && r == 9) return FFL_PYR_H9;
    return -1;
}

// (running the coarse kernel on a side stream beside the fine levels was tried: 5595 vs 5606 pairs/s, not kept)
bool ffl_launch_pyr_multi(const uint8_t *gray_base, size_t gray_stride, const UTab *__restrict__ ut, int nU, int w, int h, const PyrJob *lv,
                          int n, const FflOptions &opt, hipStream_t st) {
    if (n > FFL_MAX_JOBS) return false;
    PyrJobs A = {}, B = {};
    unsigned ta = 0, tb = 0;
    for (int i = 0; i < n; i++)
        if (ffl_pyr_kind(w, h, lv[i].lw, lv[i].lh, lv[i].gk.ksize) < 0) return false;
    // the x1/8 and x1/4 levels (radius 9 and 4) in one pass over the frame where the sizes allow it
    int skip0 = -1, skip1 = -1;
    if (opt.pyr_coarse)
        for (int i = 0; i + 1 < n; i++)
            if (lv[i].gk.ksize == 19 && lv[i + 1].gk.ksize == 9 &&
                ffl_launch_pyr_coarse(gray_base, gray_stride, ut, nU, w, h, lv[i + 1], lv[i], st)) {
                skip0 = i;
                skip1 = i + 1;
                break;
            }
    for (int i = 0; i < n; i++) {
        if (i == skip0 || i == skip1) continue;
        const int kind = ffl_pyr_kind(w, h, lv[i].lw, lv[i].lh, lv[i].gk.ksize);
        PyrJob J = lv[i];
        J.w = w;
        J.h = h;
        J.sx = (double)w / J.lw;
        J.sy = (double)h / J.lh;
        J.kind = kind;
        if (kind == FFL_PYR_F1 || kind == FFL_PYR_F2) {
            const int fr = kind == FFL_PYR_F1 ? FFL_FR1 : FFL_FR2;
            J.gx = (J.lw / 4 + 255) / 256;
            J.gy = (J.lh + fr - 1) / fr;
        } else {
            const int rows = ffl_pyr_rows(kind == FFL_PYR_H4 ? 4 : 9);
            J.gx = (J.lw + 255) / 256;
            J.gy = (h + rows - 1) / rows;
            PyrJob V = J;
            V.kind = kind == FFL_PYR_H4 ? FFL_PYR_V4 : FFL_PYR_V9;
            V.gx = (J.lw + 63) / 64;
            V.gy = (J.lh + 3) / 4;
            V.first = tb;
            V.count = V.gx * V.gy * (unsigned)nU;
            tb += ffl_xcd_blocks(V.count);
            B.j[B.n++] = V;
        }
        J.first = ta;
        J.count = J.gx * J.gy * (unsigned)nU;
        ta += ffl_xcd_blocks(J.count);
        A.j[A.n++] = J;
    }
    if (A.n) hipLaunchKernelGGL(k_pyr_multi, dim3(ta), dim3(256), 0, st, gray_base, gray_stride, ut, A);
    if (B.n) hipLaunchKernelGGL(k_pyr_multi, dim3(tb), dim3(256), 0, st, gray_base, gray_stride, ut, B);
    return true;
}

void ffl_launch_pyr_level(const uint8_t *gray_base, size_t gray_stride, const UTab *__restrict__ ut, int nU, int w, int h, int lw, int lh,
                          GaussKernel gk, float *tmp, size_t tmp_stride, float *I, size_t I_stride, hipStream_t st) {
    const int r = gk.ksize / 2, nq = lw != w ? 2 : 1;
    const double sx = (double)w / lw, sy = (double)h / lh;
    if (r == 1 && ((lw == w && lh == h) || (w == 2 * lw && h == 2 * lh)) && (lw & 3) == 0 && (w & 3) == 0) {
        constexpr int FR1 = FFL_FR1, FR2 = FFL_FR2;  // output rows per lane
        if (lw == w) {
            dim3 grid((lw / 4 + 255) / 256, (lh + FR1 - 1) / FR1, nU);
            hipLaunchKernelGGL((k_pyr_fused3x4<1, FR1>), grid, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, lh,
                               gk, I, I_stride);
        } else {
            dim3 grid((lw / 4 + 255) / 256, (lh + FR2 - 1) / FR2, nU);
            hipLaunchKernelGGL((k_pyr_fused3x4<2, FR2>), grid, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, lh,
                               gk, I, I_stride);
        }
        return;
    }
    if (r == 1 && ((lw == w && lh == h) || (w == 2 * lw && h == 2 * lh))) {
        dim3 grid((lw + 255) / 256, (lh + FFL_PYR_FROWS - 1) / FFL_PYR_FROWS, nU);
        if (lw == w)
            hipLaunchKernelGGL(k_pyr_fused3<1>, grid, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, lh, gk, I,
                               I_stride);
        else
            hipLaunchKernelGGL(k_pyr_fused3<2>, grid, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, lh, gk, I,
                               I_stride);
        return;
    }
#define FFL_PYR_LAUNCH2(RR)                                                                                         \
    do {                                                                                                            \
        dim3 gv((lw + 63) / 64, (lh + 3) / 4, nU);                                                                  \
        dim3 gh((lw + 255) / 256, (h + ffl_pyr_rows(RR) - 1) / ffl_pyr_rows(RR), nU);                               \
        hipLaunchKernelGGL(k_pyr_h2<RR>, gh, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, sx, gk, tmp,   \
                           tmp_stride);                                                                             \
        hipLaunchKernelGGL(k_pyr_v2<RR>, gv, dim3(256), 0, st, tmp, tmp_stride, w, h, lw, lh, sx, sy, gk, I,        \
                           I_stride);                                                                               \
    } while (0)
    if (nq == 2 && r == 4) { FFL_PYR_LAUNCH2(4); return; }
    if (nq == 2 && r == 9) { FFL_PYR_LAUNCH2(9); return; }
#undef FFL_PYR_LAUNCH2
#define FFL_PYR_LAUNCH(RR)                                                                                          \
    do {                                                                                                            \
        dim3 gv((lw + 63) / 64, (lh + 4 * ffl_pyr_vrows(RR) - 1) / (4 * ffl_pyr_vrows(RR)), nU);                    \
        dim3 gh((lw * nq + 255) / 256, (h + ffl_pyr_rows(RR) - 1) / ffl_pyr_rows(RR), nU);                          \
        hipLaunchKernelGGL(k_pyr_h<RR>, gh, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, nq, sx, gk, tmp, \
                           tmp_stride);                                                                             \
        hipLaunchKernelGGL(k_pyr_v<RR>, gv, dim3(256), 0, st, tmp, tmp_stride, w, h, lw, lh, nq, sx, sy, gk, I,      \
                           I_stride);                                                                               \
    } while (0)
    if (r == 1) FFL_PYR_LAUNCH(1);
    else if (r == 4) FFL_PYR_LAUNCH(4);
    else if (r == 9) FFL_PYR_LAUNCH(9);
    else FFL_PYR_LAUNCH(0);
#undef FFL_PYR_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// K2: polynomial expansion (FarnebackPolyExp, n = 5, sigma = 1.2): I (lh,lw) -> R 5 planes.
// One 64x16 output tile per 256-thread workgroup; the (64+10)x(16+10) input tile and the three
// vertically filtered rows live in LDS; the horizontal pass accumulates in double.
// ------------------------------------------------------------------------------------------------
#ifndef PE_TW
#define PE_TW 64
#define PE_TH 16
#endif
#define PE_N FFL_POLY_N
#define PE_LW (PE_TW + 2 * PE_N)   // 74 tile columns incl. the halo
#define PE_LQ ((PE_LW + 3) / 4)    // 19 column quads
#define PE_PITCH (4 * PE_LQ)       // 76: rows are whole quads, every quad 16-byte aligned

// Four horizontally adjacent pixels per lane in every phase.  The kernel is bound by instruction issue (f32 taps,
// f64 accumulators, LDS reads), not by memory: with four pixels per lane the 14 taps a lane needs per plane come with
// four LDS reads (3 x 16 B + 8 B) instead of the 22 dword reads two pixels needed, the tile comes in with 16-byte
// global loads, and results leave with 16-byte stores.  Per pixel the operations and their order are unchanged.
__device__ __forceinline__ void k_polyexp_body(const unsigned bx, const unsigned by, const unsigned bz, const float *__restrict__ I, size_t I_stride, float *__restrict__ R,
                                                 size_t R_stride, size_t plane, int w, int h, PolyConsts pc) {
    __shared__ __attribute__((aligned(16))) float sI[PE_TH + 2 * PE_N][PE_PITCH];
    __shared__ __attribute__((aligned(16))) float sV[3][PE_TH][PE_PITCH];
    const int tid = threadIdx.x;
    const int u = bz;
    const int x0 = bx * PE_TW, y0 = by * PE_TH;
    const float *img = I + (size_t)u * I_stride;

    // fixed trip counts (+ a bounds predicate) so that the loops unroll: a rolled loop issues one
    // global load, waits for it, stores it to LDS, and only then issues the next
    constexpr int N_IN = (PE_TH + 2 * PE_N) * PE_LQ, N_V = PE_TH * PE_LQ;
#pragma unroll
    for (int it = 0; it < (N_IN + 255) / 256; it++) {
        const int i = tid + 256 * it;
        if (i < N_IN) {
            const int ly = i / PE_LQ, lx = 4 * (i - ly * PE_LQ);
            const int gy = min(max(y0 + ly - PE_N, 0), h - 1), gx = x0 + lx - PE_N;
            const float *row = img + (size_t)gy * w;
            float4 t;
            if (gx >= 0 && gx + 3 < w) {
                const ffl_f4u q = *reinterpret_cast<const ffl_f4u *>(row + gx);
                t = make_float4(q.x, q.y, q.z, q.w);
            } else {  // REPLICATE border
                t = make_float4(row[min(max(gx, 0), w - 1)], row[min(max(gx + 1, 0), w - 1)],
                                row[min(max(gx + 2, 0), w - 1)], row[min(max(gx + 3, 0), w - 1)]);
            }
            *reinterpret_cast<float4 *>(&sI[ly][lx]) = t;
        }
    }
    __syncthreads();

    // vertical part (float): rows are clamped because the tile was loaded with clamped rows
#pragma unroll
    for (int it = 0; it < (N_V + 255) / 256; it++) {
        const int i = tid + 256 * it;
        if (i >= N_V) break;
        const int ly = i / PE_LQ, lx = 4 * (i - ly * PE_LQ);
        const float4 c4 = *reinterpret_cast<const float4 *>(&sI[ly + PE_N][lx]);
        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
        float r0[4], r1[4], r2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            r0[e] = c[e] * pc.g[0];
            r1[e] = 0.f;
            r2[e] = 0.f;
        }
#pragma unroll
        for (int k = 1; k <= PE_N; k++) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&sI[ly + PE_N - k][lx]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&sI[ly + PE_N + k][lx]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float p = a[e] + b[e], d = b[e] - a[e];
                r0[e] = r0[e] + pc.g[k] * p;
                r1[e] = r1[e] + pc.xg[k] * d;
                r2[e] = r2[e] + pc.xxg[k] * p;
            }
        }
        *reinterpret_cast<float4 *>(&sV[0][ly][lx]) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        *reinterpret_cast<float4 *>(&sV[1][ly][lx]) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        *reinterpret_cast<float4 *>(&sV[2][ly][lx]) = make_float4(r2[0], r2[1], r2[2], r2[3]);
    }
    __syncthreads();

    // horizontal part (double accumulators; the b2,b3,b5,b6 products are float products): one item = 4 pixels,
    // 16 lanes per tile row, 256 items = one per lane
    float *out = R + (size_t)u * R_stride;
    {
        static_assert(PE_TW * PE_TH == 1024 && PE_TW % 4 == 0, "one 4-pixel item per lane");
        const int ly = tid / (PE_TW / 4), lx = 4 * (tid % (PE_TW / 4));
        const int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) return;
        float t[3][16];  // taps: tile columns lx .. lx+13 of the three planes (pixel e is centred at lx + 5 + e)
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const float *src = &sV[p][ly][lx];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
                t[p][4 * q] = v.x; t[p][4 * q + 1] = v.y; t[p][4 * q + 2] = v.z; t[p][4 * q + 3] = v.w;
            }
            const float2 v2 = *reinterpret_cast<const float2 *>(src + 12);
            t[p][12] = v2.x; t[p][13] = v2.y;
        }
        float o[5][4];
        const float g0c = pc.g[0];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = PE_N + e;
            double b1 = (double)(t[0][c] * g0c), b2 = 0, b3 = (double)(t[1][c] * g0c), b4 = 0, b5 = (double)(t[2][c] * g0c), b6 = 0;
#pragma unroll
            for (int k = 1; k <= PE_N; k++) {
                const float gk = pc.g[k], xgk = pc.xg[k];
                const float s0 = t[0][c + k] + t[0][c - k];
                const float d0 = (t[0][c + k] - t[0][c - k]) * xgk;
                const float s1 = (t[1][c + k] + t[1][c - k]) * gk;
                const float d1 = (t[1][c + k] - t[1][c - k]) * xgk;
                const float s2 = (t[2][c + k] + t[2][c - k]) * gk;
                // tg, gd[k], xxgd[k] are floats widened to double: their product has <= 48 significant bits, i.e. it is
                // EXACT in double, so the fused multiply-add rounds exactly what "b += tg * g" rounds -- bit-identical to the
                // oracle's separate multiply and add, two f64-rate instructions fewer per tap (this kernel is bound by them)
                const double tg = (double)s0;
                b1 = __builtin_fma(tg, pc.gd[k], b1);
                b4 = __builtin_fma(tg, pc.xxgd[k], b4);
                b2 += (double)d0;
                b3 += (double)s1;
                b6 += (double)d1;
                b5 += (double)s2;
            }
            o[0][e] = (float)(b3 * pc.ig11);
            o[1][e] = (float)(b2 * pc.ig11);
            o[2][e] = (float)(b1 * pc.ig03 + b5 * pc.ig33);
            o[3][e] = (float)(b1 * pc.ig03 + b4 * pc.ig33);
            o[4][e] = (float)(b6 * pc.ig55);
        }
        const size_t off = (size_t)y * w + x;
        if (x + 3 < w) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                ffl_f4u q;
                q.x = o[c][0]; q.y = o[c][1]; q.z = o[c][2]; q.w = o[c][3];
                *reinterpret_cast<ffl_f4u *>(out + c * plane + off) = q;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (x + e < w) out[c * plane + off + e] = o[c][e];
        }
    }
}
__global__ __launch_bounds__(256) void k_polyexp(const float *__restrict__ I, size_t I_stride, float *__restrict__ R,
                                                 size_t R_stride, size_t plane, int w, int h, PolyConsts pc) {
    k_polyexp_body(blockIdx.x, blockIdx.y, blockIdx.z, I, I_stride, R, R_stride, plane, w, h, pc);
}

// All levels' PolyExp in ONE launch (a 1-D grid cut into per-level ranges): the coarse levels' small grids
// run inside the level-0 launch instead of paying a launch ramp and tail each.
__global__ __launch_bounds__(256) void k_polyexp_multi(PolyJobs jobs, PolyConsts pc) {
    int i = 0;
#pragma unroll
    for (int t = 1; t < FFL_MAX_JOBS; t++)
        if (t < jobs.n && blockIdx.x >= jobs.j[t].first) i = t;
    const PolyJob &J = jobs.j[i];
    // Plain order on purpose.  The XCD-aware order (ffl_xcd_tile: every XCD walks its own contiguous run of tiles, so
    // the 5-pixel halo is re-read from that XCD's L2) cut this kernel's fetch from 1.09 GB to 0.36 GB per 33-frame step
    // -- 3.0x -> 1.0x of its input -- and made it 7 % SLOWER (583 -> 625 us): the kernel is bound by its 20 B/px write
    // stream, which prefers neighbouring tiles to be written at the same time by all XCDs.
    const unsigned t = blockIdx.x - J.first;
    if (t >= J.count) return;
    const unsigned per = J.gx * J.gy;
    const unsigned bz = t / per, r = t - bz * per, by = r / J.gx, bx = r - by * J.gx;
    k_polyexp_body(bx, by, bz, J.I, J.I_stride, J.R, J.R_stride, J.plane, J.w, J.h, pc);
}

void ffl_launch_polyexp_multi(const PolyJob *jobs_in, int n, int nU, PolyConsts pc, hipStream_t st) {
    PolyJobs jobs = {};
    unsigned total = 0;
    for (int i = 0; i < n; i++) {
        jobs.j[i] = jobs_in[i];
        jobs.j[i].gx = (jobs_in[i].w + PE_TW - 1) / PE_TW;
        jobs.j[i].gy = (jobs_in[i].h + PE_TH - 1) / PE_TH;
        jobs.j[i].first = total;
        jobs.j[i].count = jobs.j[i].gx * jobs.j[i].gy * (unsigned)nU;
        total += jobs.j[i].count;
    }
    jobs.n = n;
    hipLaunchKernelGGL(k_polyexp_multi, dim3(total), dim3(256), 0, st, jobs, pc);
}

void ffl_launch_polyexp(const float *I, size_t I_stride, float *R, size_t R_stride, size_t plane, int nU, int lw,
                        int lh, PolyConsts pc, hipStream_t st) {
    dim3 grid((lw + PE_TW - 1) / PE_TW, (lh + PE_TH - 1) / PE_TH, nU);
    hipLaunchKernelGGL(k_polyexp, grid, dim3(256), 0, st, I, I_stride, R, R_stride, plane, lw, lh, pc);
}

// ------------------------------------------------------------------------------------------------
// K3 + K4: FarnebackUpdateMatrices (standalone; runs once per level before the first blur
// iteration).  With UPSAMPLE the level's initial flow -- K3: resize(prevFlow, (w, h), INTER_LINEAR)
// * (1 / pyrScale) -- is produced here as well, written once and used from registers.  64x16 tiles
// in XCD-aware panel order: the R1 rows y1, y1+1 gathered by vertically adjacent tiles are re-read
// from L2.  usx / usy = (double)pw / w, (double)ph / h.
// ------------------------------------------------------------------------------------------------
// MODE 0: the flow field is read; 1: it is formed here (x2 upsample of the coarser level) and written; 2: it is
// zero (coarsest level) and neither read nor written -- k_blur_solve overwrites it without reading it
template <int MODE>
__global__ __launch_bounds__(256) void k_update_matrices(const float *__restrict__ R, size_t R_stride, size_t plane,
                                                         const PairTab *__restrict__ pt, int level,
                                                         float *__restrict__ M, size_t M_stride, int w,
                                                         int h, int pw, int ph, double usx, double usy, int store_flow,
                                                         int nB, int order) {
    // One pixel per lane, lanes along x (a wave = one 64-pixel tile row; wave q takes rows q, q+4, q+8, q+12), as the
    // fused update of k_blur_solve: all four rows' flow vectors first, then R0 + the R1 corners of the lane's NEXT row
    // are requested before the current row's arithmetic.  Dword loads for R0 / R1, 16-byte loads for the coarse flow
    // (a wave64 8-byte load costs the vector-memory path three dword loads: profiles/tools/micro/vmem_issue.hip).
    int b, tile_x, tile_y;
    if (!ffl_tile_coord((w + 63) / 64, (h + 15) / 16, nB, order, b, tile_x, tile_y)) return;
    const int lx = threadIdx.x & 63, x0 = tile_x * 64, x = min(x0 + lx, w - 1);
    const bool xin = x0 + lx < w;
    const float *R0 = R + (size_t)pt->u0[b] * R_stride, *R1 = R + (size_t)pt->u1[b] * R_stride;
    float *flow = pt->flow[level][b];
    const float *pf = pt->flow[min(level + 1, FFL_MAX_LEVELS - 1)][b];
    float *Mb = M + (size_t)b * M_stride;
    constexpr int NR = 4;
    int yr[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) yr[k] = tile_y * 16 + (threadIdx.x >> 6) + 4 * k;
    float2 f[NR];
    if (MODE == 1) {
        const bool half_scale = usx == 0.5 && usy == 0.5;
        int xa0, xa1;
        float a1;
        if (half_scale) ffl_resize_coord_half(x, pw, xa0, xa1, a1);
        else ffl_resize_coord(x, pw, usx, xa0, xa1, a1);
        const int xq = min(xa0, pw - 2);   // columns xq, xq + 1 lie inside the row; xa0 and xa1 are each one of them
        const bool first0 = xa0 == xq, first1 = xa1 == xq;
        const float a0 = 1.f - a1;
        float4 c0[NR], c1[NR];
        float b1[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            int ya0, ya1;
            const int y = min(yr[k], h - 1);
            if (half_scale) ffl_resize_coord_half(y, ph, ya0, ya1, b1[k]);
            else ffl_resize_coord(y, ph, usy, ya0, ya1, b1[k]);
            c0[k] = ffl_gload4(pf, 8u * ((unsigned)ya0 * (unsigned)pw + xq));
            c1[k] = ffl_gload4(pf, 8u * ((unsigned)ya1 * (unsigned)pw + xq));
        }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const float b0 = 1.f - b1[k];
            const float2 p00 = first0 ? make_float2(c0[k].x, c0[k].y) : make_float2(c0[k].z, c0[k].w);
            const float2 p01 = first1 ? make_float2(c0[k].x, c0[k].y) : make_float2(c0[k].z, c0[k].w);
            const float2 p10 = first0 ? make_float2(c1[k].x, c1[k].y) : make_float2(c1[k].z, c1[k].w);
            const float2 p11 = first1 ? make_float2(c1[k].x, c1[k].y) : make_float2(c1[k].z, c1[k].w);
            float t0 = p00.x * a0 + p01.x * a1, t1 = p10.x * a0 + p11.x * a1;
            f[k].x = (t0 * b0 + t1 * b1[k]) * 2.0f;
            t0 = p00.y * a0 + p01.y * a1;
            t1 = p10.y * a0 + p11.y * a1;
            f[k].y = (t0 * b0 + t1 * b1[k]) * 2.0f;
        }
    } else if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const unsigned o = (unsigned)min(yr[k], h - 1) * (unsigned)w + (unsigned)x;
            const FFL_GLOBAL float *q = (const FFL_GLOBAL float *)flow + 2u * o;
            f[k] = make_float2(q[0], q[1]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NR; k++) f[k] = make_float2(0.f, 0.f);
    }
    struct UmRow {
        float a00, a01, a10, a11;
        bool inside;
        float r0[5];
        ffl_f2u t[5], u[5];
    };
    auto issue = [&](int k) {
        UmRow S;
        const int y = min(yr[k], h - 1);
        const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
#pragma unroll
        for (int c = 0; c < 5; c++) S.r0[c] = *ffl_at<float>(R0 + c * plane, o);
        const UmLoc L = ffl_um_locate(w, h, x, y, f[k].x, f[k].y);
        S.a00 = L.a00; S.a01 = L.a01; S.a10 = L.a10; S.a11 = L.a11; S.inside = L.inside;
        const unsigned o1 = L.inside ? (unsigned)L.y1 * (unsigned)w + (unsigned)L.x1 : 0u;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            S.t[c] = ffl_ld_corner(R1 + c * plane, o1);
            S.u[c] = ffl_ld_corner(R1 + c * plane, o1 + (unsigned)w);
        }
        return S;
    };
    UmRow cur = issue(0);
#pragma unroll
    for (int k = 0; k < NR; k++) {
        UmRow nxt;
        if (k + 1 < NR) nxt = issue(k + 1);
        const bool in = xin && yr[k] < h;
        const int y = min(yr[k], h - 1);
        const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
        // the upsampled field is consumed right here; nothing downstream reads it (k_blur_solve overwrites the flow
        // without reading it), so it only goes to memory for the debug capture
        if (MODE == 1 && in && store_flow) {
            flow[2u * o] = f[k].x;
            flow[2u * o + 1u] = f[k].y;
        }
        float bb[5];
#pragma unroll
        for (int c = 0; c < 5; c++) bb[c] = cur.a00 * cur.t[c].x + cur.a01 * cur.t[c].y + cur.a10 * cur.u[c].x + cur.a11 * cur.u[c].y;
        float m[5];
        ffl_um_finish(cur.r0, bb, cur.inside, w, h, x, y, f[k].x, f[k].y, m);
        if (in) {
#pragma unroll
            for (int c = 0; c < 5; c++) *ffl_at<float>(Mb + c * plane, o) = m[c];
        }
        if (k + 1 < NR) cur = nxt;
    }
}

void ffl_launch_update_matrices(const float *R, size_t R_stride, size_t plane, const PairTab *pt, int level, int nB,
                                float *M,
                                size_t M_stride, int lw, int lh, int pw, int ph, int zero_flow, int store_flow,
                                const FflOptions &opt, hipStream_t st) {
    dim3 grid(ffl_tile_grid((lw + 63) / 64, (lh + 15) / 16, nB));
    if (pw > 0)
        hipLaunchKernelGGL(k_update_matrices<1>, grid, dim3(256), 0, st, R, R_stride, plane, pt, level, M, M_stride, lw, lh,
                           pw, ph, (double)pw / lw, (double)ph / lh, store_flow, nB, opt.tile_order);
    else if (zero_flow)
        hipLaunchKernelGGL(k_update_matrices<2>, grid, dim3(256), 0, st, R, R_stride, plane, pt, level, M, M_stride, lw, lh, 0, 0,
                           1.0, 1.0, 0, nB, opt.tile_order);
    else
        hipLaunchKernelGGL(k_update_matrices<0>, grid, dim3(256), 0, st, R, R_stride, plane, pt, level, M, M_stride, lw, lh, 0, 0,
                           1.0, 1.0, 0, nB, opt.tile_order);
}

// ------------------------------------------------------------------------------------------------
// K5: FarnebackUpdateFlow_Blur: 15x15 box sum of the 5 M planes (double, REPLICATE border, rows
// then columns, every 15-term sum in the position-anchored block order of the oracle's
// box15_block16()), 2x2 solve in double, flow write, and -- when UPDATE -- the next UpdateMatrices
// fused on the freshly solved displacement (written to the other M buffer, so neighbouring tiles
// still read the old M: a Jacobi step, identical to OpenCV's striped in-place update).
//
// Summation order (what makes the sums cheap AND bit-identical to the oracle): a column / row is cut
// into blocks of 16 anchored at 16k-8; the window of position 16k+t is (suffix of block k from t+1,
// accumulated backwards from the block's end) + (prefix of block k+1 up to t-1, accumulated forwards).
// The 64x16 tiles are aligned to those blocks, so a lane that owns a whole block column forms its 16
// sums with 42 additions, and a lane that owns 4 pixels of a block row needs 20.
//
// One 64 x 16 output tile per 256-thread workgroup:
//   phase V  390 lanes = 5 channels x 78 tile columns (two groups, 3 + 2 channels); each lane reads
//            its 30 rows straight from global memory (consecutive lanes = consecutive x: coalesced),
//            forms the 16 column sums and writes them to LDS as doubles;
//   phase H  each lane owns 4 consecutive pixels of one row, wave q of the workgroup the q-th quarter
//            of every 16-block (so the order of its additions is wave-uniform): 9 ds_read_b128 per
//            channel bring the 18 column sums it needs;
//   solve    2x2 system in double, the tile's displacements transposed through LDS; the level's last iteration
//            stores the field (two pixels per lane, 16-byte stores), the updating iterations run the fused
//            UpdateMatrices one pixel per lane with DWORD R0 loads / R1 corner gathers / M stores (a wave64 8-byte
//            access costs the vector-memory path three dword accesses).
// ------------------------------------------------------------------------------------------------
#ifndef FFL_K5_WAVES
#define FFL_K5_WAVES 4  // waves per SIMD the register allocator must leave room for (= workgroups per CU)
#endif
#ifndef FFL_K5_WAVES_FIRST
#define FFL_K5_WAVES_FIRST 3
#endif
#ifndef FFL_K5_GROUP_FIRST
#define FFL_K5_GROUP_FIRST 3  // the same for the folded first iteration (2 -> 38 KB of LDS: 4 workgroups per CU)
#endif
#ifndef FFL_K5_GROUP
#define FFL_K5_GROUP 3  // channels that go through LDS together (3 -> 3+2, 2 -> 2+2+1)
#endif

// v[0..29] = positions 16k-7 .. 16k+22 of a column (float, widened at use); out[t] = window of 16k+t
__device__ __forceinline__ void ffl_box_block16(const float (&v)[30], double (&out)[16]) {
    double s[15];
    s[14] = (double)v[14];
#pragma unroll
    for (int j = 13; j >= 0; j--) s[j] = (double)v[j] + s[j + 1];
    double p = (double)v[15];
    out[0] = s[0];
#pragma unroll
    for (int t = 1; t < 15; t++) {
        out[t] = s[t] + p;
        p = p + (double)v[15 + t];
    }
    out[15] = p;
}

// The same sums for the 4 pixels t = 4Q .. 4Q+3 of a block row: d[0..17] = column sums of positions
// 16k+4Q-7 .. 16k+4Q+10, of which d[0..NA-1] lie in block k and the rest in block k+1.
template <int Q>
__device__ __forceinline__ void ffl_box_quarter(const double (&d)[18], double (&out)[4]) {
    constexpr int NA = 15 - 4 * Q;
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    double s = d[NA - 1];
    if (NA - 1 < 4) S[NA - 1] = s;
#pragma unroll
    for (int j = NA - 2; j >= 0; j--) {
        s = d[j] + s;
        if (j < 4) S[j] = s;
    }
    double p = d[NA];  // prefix of block k+1 up to offset m: P(m)
#pragma unroll
    for (int m = 0; m <= 4 * Q + 2; m++) {
        if (m > 0) p = p + d[NA + m];
        const int i = m + 1 - 4 * Q;  // pixel t = 4Q+i takes P(t-1)
        if (i >= 0 && i < 4) {
            if (4 * Q + i == 15) out[i] = p;
            else out[i] = S[i] + p;
        }
    }
    if (Q == 0) out[0] = S[0];  // t = 0: the window is the suffix alone
}

#ifdef FFL_STAMP
// Diagnostic build only (profiles/tools/k5_stamps.sh): thread 0 of every workgroup of the level-0 folded launch adds the
// cycles between consecutive stamp points into LDS counters and leaves them in a device array the host reads back.
// Shares, not absolute times, are what this build is good for (cdna_hip_programming.md section 7, in-kernel stamps).
#define FFL_NSTAMP 16
__device__ unsigned long long g_stamp[64][FFL_NSTAMP];
extern "C" int ffl_debug_read_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 64 * FFL_NSTAMP);
}
#define FFL_T(i)                                                              \
    if (FFL_STAMP_ON && tid == 0) {                                           \
        const unsigned long long t_ = __builtin_readcyclecounter();          \
        sStamp[i] += t_ - sStamp[FFL_NSTAMP];                                 \
        sStamp[FFL_NSTAMP] = t_;                                              \
    }
#else
#define FFL_T(i)
#endif


// FIRST != 0: the level's first iteration with the initial UpdateMatrices folded in -- M is never read from
// memory: a "phase U" computes it for the 78 tile columns x 16 new rows from R0, R1 and the level's initial
// flow (FIRST == 1: the x2 upsample of the coarser level's flow, formed on the fly and never stored;
// FIRST == 2: zero) into LDS, where phase V picks it up.  Thanks to the strip walk only the 14 halo columns
// are computed redundantly (x1.22), and the level saves one whole M write + read (40 B per pixel) and a launch.
template <bool UPDATE, int FIRST>
__global__ __launch_bounds__(256, FIRST ? FFL_K5_WAVES_FIRST : FFL_K5_WAVES) void k_blur_solve(
    const float *__restrict__ Min, float *__restrict__ Mout, size_t M_stride, const float *__restrict__ R, size_t R_stride,
    size_t plane, const PairTab *__restrict__ pt, int level, int w, int h, int nrb, int pw, int ph, double usx, double usy,
    int store_flow, int nB,
    int order) {
    constexpr int TW = 64, TH = 16, LW = TW + 2 * FFL_WIN_R, LH = TH + 2 * FFL_WIN_R;
    constexpr int PX = 4;  // consecutive pixels per lane in phase H
    constexpr int NCARRY = LH - TH;  // rows of a tile's 30 that the tile below needs again
    // The 5 channels go through LDS in two groups (3 + 2): 3*16*88 doubles = 33 KB, so four workgroups
    // fit a CU, and each group's 3*78 / 2*78 column lanes fit one pass of 256 lanes.  Column tx of a row
    // sits at double index tx + 2*(tx >> 4): one 16-byte pad per 16 columns and a row pitch of 44 x 16 B
    // make the 64 ds_read_b128 of a phase-H wave (16 rows x 4 blocks, 144 B apart) hit 8 distinct
    // 16-byte bank groups per 8 lanes.
    constexpr int LW2 = 44;
    constexpr int GC = FIRST ? FFL_K5_GROUP_FIRST : FFL_K5_GROUP, NG = (5 + GC - 1) / GC;  // channels per LDS pass, passes
    __shared__ double2 sS2[GC][TH][LW2];
    const int tid = threadIdx.x;
#ifdef FFL_STAMP
    __shared__ unsigned long long sStamp[FFL_NSTAMP + 1];
    const bool FFL_STAMP_ON = FIRST == 1 && w >= 1024;
    if (FFL_STAMP_ON && tid == 0) {
        for (int i = 0; i < FFL_NSTAMP; i++) sStamp[i] = 0;
        sStamp[FFL_NSTAMP] = __builtin_readcyclecounter();
    }
#endif
    // A workgroup walks down `nrb` vertically adjacent tiles (a column strip of 64 x 16*nrb pixels) and
    // keeps the 14 rows two consecutive tiles share in registers: phase V then loads 16 new rows per tile
    // instead of 30 -- its halo re-reads (30 rows for 16 outputs) were the kernel's largest single cost.
    int b, tile_x, strip_y;  // XCD-aware panel order over the strips, see ffl_tile_coord
    const int tiles_y = (h + TH - 1) / TH;
    if (!ffl_tile_coord((w + TW - 1) / TW, (tiles_y + nrb - 1) / nrb, nB, order, b, tile_x, strip_y)) return;
    const int x0 = tile_x * TW;
    const float *Mb = Min + (size_t)b * M_stride;

    // phase H / solve: wave q owns quarter q of each 16-block; inside the wave 16 rows x 4 blocks
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = (tid & 63) >> 2, kk = tid & 3;
    const int lx0 = 16 * kk + 4 * q;  // first of the lane's 4 pixels (tile coordinates)
    double *sS = reinterpret_cast<double *>(&sS2[0][0][0]);
    // phase V: (channel of the group, tile column) of this lane; the same for both groups
    const int vcc = tid / LW, vtx = tid - vcc * LW;
    const int vgx = min(max(x0 + vtx - FFL_WIN_R, 0), w - 1);
    const unsigned pitch = (unsigned)w * 4u;
    float carry[NG][NCARRY];
    const double scale = 1.0 / (FFL_WIN * FFL_WIN);
    float2 *flow = reinterpret_cast<float2 *>(pt->flow[level][b]);
    const float *R0 = R + (size_t)pt->u0[b] * R_stride, *R1 = R + (size_t)pt->u1[b] * R_stride;
    float *Mo = Mout + (size_t)b * M_stride;
    // float2 sF[TH][FP] viewed as float4 pairs; row pitch 66 (not 64) float2 so that the 64 32-byte writes
    // of a wave (16 rows x 4 blocks, all at the same offset inside their 128-byte block) spread over the banks
    constexpr int FP = TW + 2;
    float4 *sF4 = reinterpret_cast<float4 *>(&sS2[0][0][0]);
    // phase U staging: M of channels 0..2 aliases the column-sum buffer (consumed before the sums are written),
    // channels 3, 4 wait in their own 10 KB until the second channel group's phase V
    constexpr int UP = 80;  // row pitch (floats) of the staged M rows
    __shared__ float sM34[FIRST ? 5 - GC : 1][FIRST ? TH : 1][FIRST ? UP : 1];  // channels GC .. 4
    float *sM012 = reinterpret_cast<float *>(&sS2[0][0][0]);                     // channels 0 .. GC-1: [GC][TH][UP]
    static_assert(GC * TH * UP * 4 <= (int)sizeof(sS2), "the first group's staged rows must fit the column sums");
    const float2 *prevf = reinterpret_cast<const float2 *>(pt->flow[min(level + 1, FFL_MAX_LEVELS - 1)][b]);
    // phase U: M of the tile rows whose phase-V index is jbase .. jbase+nrows-1 (image rows y0-7+jbase ..), all 78
    // columns; the level's initial flow is the x2 upsample of the coarser level's (K3: resize(prevFlow, (w, h),
    // INTER_LINEAR) * 2, as in k_update_matrices<1>), formed per pixel and never stored.
    // Software-pipelined phase U, one pixel per lane.  An item costs two DEPENDENT memory round trips (the coarse
    // flow's corners, then R1 at the displaced position); with 3 waves per SIMD the waves sat in s_waitcnt half of
    // their cycles.  The NEXT item's flow corners are requested right after the current item's gathers, so that
    // round trip runs under the gathers' wait and the item's arithmetic.  One pixel per lane (not the pair of the
    // standalone kernel) keeps the two items in flight within the 168-register budget of 3 waves per SIMD: the pair
    // version of the same pipeline spilled 55 registers; one stage deeper (the next item's R1 gathers in flight as
    // well) spilled too.  Load widths follow the cost of a wave64 global load on the CU's vector-memory path
    // (profiles/tools/micro/vmem_issue.hip: 4 / 8 / 16 bytes per lane = 6.3 / 19.5 / 16.7 cycles): dwords for R0 and
    // the R1 corners, one 16-byte load per coarse-flow row, never 8 bytes.  DESIGN.md section 4, round 2 (v), (vi).
    struct UStageA {
        float a1, b1;
        float4 c0, c1;      // rows ya0 / ya1 of the coarse flow, columns xq and xq + 1 (one 16-byte load each)
        bool first0, first1;  // column xa0 / xa1 is the first of the two
        int r, tx, gx, gy;    // the item's coordinates, formed once
    };
    const bool half_scale = usx == 0.5 && usy == 0.5;
    auto phase_u = [&](int y0, int jbase, int nrows) {
        const int N = LW * nrows;
        auto coords = [&](int i, int &r, int &tx, int &gx, int &gy) {
            r = i / LW;
            tx = i - r * LW;
            gy = min(max(y0 - FFL_WIN_R + jbase + r, 0), h - 1);
            gx = min(max(x0 + tx - FFL_WIN_R, 0), w - 1);
        };
        auto issue = [&](int i) {
            UStageA A;
            coords(i, A.r, A.tx, A.gx, A.gy);
            const int gx = A.gx, gy = A.gy;
            if (FIRST == 1) {
                int xa0, xa1, ya0, ya1;
                if (half_scale) {  // uniform: the usual case (even level sizes) without the f64 coordinate arithmetic
                    ffl_resize_coord_half(gx, pw, xa0, xa1, A.a1);
                    ffl_resize_coord_half(gy, ph, ya0, ya1, A.b1);
                } else {
                    ffl_resize_coord(gx, pw, usx, xa0, xa1, A.a1);
                    ffl_resize_coord(gy, ph, usy, ya0, ya1, A.b1);
                }
                const float *pf = reinterpret_cast<const float *>(prevf);
                const unsigned r0o = (unsigned)ya0 * (unsigned)pw, r1o = (unsigned)ya1 * (unsigned)pw;
                // global-address loads (ffl_gload4): behind a FLAT load the compiler can only wait for "everything".
                // The two x-neighbours of a row with ONE 16-byte load (a wave64 dwordx4 costs the vector-memory path less
                // than a dwordx2): columns xq, xq + 1 with xq = min(xa0, pw - 2) always lie inside the row, and xa0, xa1
                // are each one of them (xa1 = xa0 + 1, or = xa0 at the right border); the choice is made at the use
                const int xq = min(xa0, pw - 2);
                A.c0 = ffl_gload4(pf, 8u * (r0o + xq));
                A.c1 = ffl_gload4(pf, 8u * (r1o + xq));
                A.first0 = xa0 == xq;
                A.first1 = xa1 == xq;
            }
            return A;
        };
        int i = tid;
        if (i >= N) return;
        UStageA A = issue(i);
#pragma unroll 1
        for (;;) {
            const int r = A.r, tx = A.tx, gx = A.gx, gy = A.gy;
            const int inext = i + 256;
            const bool more = inext < N;
            float2 f = make_float2(0.f, 0.f);
            if (FIRST == 1) {
                const float a0 = 1.f - A.a1, b0 = 1.f - A.b1;
                const float2 p00 = A.first0 ? make_float2(A.c0.x, A.c0.y) : make_float2(A.c0.z, A.c0.w);
                const float2 p01 = A.first1 ? make_float2(A.c0.x, A.c0.y) : make_float2(A.c0.z, A.c0.w);
                const float2 p10 = A.first0 ? make_float2(A.c1.x, A.c1.y) : make_float2(A.c1.z, A.c1.w);
                const float2 p11 = A.first1 ? make_float2(A.c1.x, A.c1.y) : make_float2(A.c1.z, A.c1.w);
                float t0 = p00.x * a0 + p01.x * A.a1, t1 = p10.x * a0 + p11.x * A.a1;
                f.x = (t0 * b0 + t1 * A.b1) * 2.0f;
                t0 = p00.y * a0 + p01.y * A.a1;
                t1 = p10.y * a0 + p11.y * A.a1;
                f.y = (t0 * b0 + t1 * A.b1) * 2.0f;
            }
            float r0[5];
            {
                const unsigned o = (unsigned)gy * (unsigned)w + (unsigned)gx;
#pragma unroll
                for (int c = 0; c < 5; c++) r0[c] = *ffl_at<float>(R0 + c * plane, o);
            }
            const UmLoc L = ffl_um_locate(w, h, gx, gy, f.x, f.y);
            // branch-free gather: lanes that land outside read the (valid) corner of pixel (0, 0) and drop it
            const unsigned o1 = L.inside ? (unsigned)L.y1 * (unsigned)w + (unsigned)L.x1 : 0u;
            ffl_f2u t[5], u[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                t[c] = ffl_ld_corner(R1 + c * plane, o1);
                u[c] = ffl_ld_corner(R1 + c * plane, o1 + (unsigned)w);
            }
            // unconditional (a lane without a next item re-requests its own corners, cache hits): a fixed number of
            // loads behind the gathers lets the wait below be "all but the last 2", not "all"
            const UStageA Anext = issue(more ? inext : i);
            float b[5];
#pragma unroll
            for (int c = 0; c < 5; c++) b[c] = L.a00 * t[c].x + L.a01 * t[c].y + L.a10 * u[c].x + L.a11 * u[c].y;
            float m[5];
            ffl_um_finish(r0, b, L.inside, w, h, gx, gy, f.x, f.y, m);
#pragma unroll
            for (int c = 0; c < GC; c++) sM012[(c * TH + r) * UP + tx] = m[c];
#pragma unroll
            for (int c = GC; c < 5; c++) sM34[c - GC][r][tx] = m[c];
            if (!more) break;
            A = Anext;
            i = inext;
        }
    };

#pragma unroll 1
    for (int rb = 0; rb < nrb; rb++) {
        const int y0 = (strip_y * nrb + rb) * TH;
        if (y0 >= h) break;
        double acc[5][PX];
        if (FIRST) {
            if (rb == 0) {  // the strip's first tile: its 14 upper halo rows go straight into the carry registers
                phase_u(y0, 0, NCARRY);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int c0 = g * GC, nc = min(GC, 5 - c0);
                    if (tid < nc * LW) {
#pragma unroll
                        for (int j = 0; j < NCARRY; j++)
                            carry[g][j] = g == 0 ? sM012[(vcc * TH + j) * UP + vtx] : sM34[c0 - GC + vcc][j][vtx];
                    }
                }
                __syncthreads();
            }
            FFL_T(0)   // loop overhead / previous tile's end barrier already counted
            phase_u(y0, NCARRY, TH);
            FFL_T(1)   // phase U work
            __syncthreads();
            FFL_T(2)   // wait after U
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int c0 = g * GC, nc = min(GC, 5 - c0);
            // ---- phase V: column sums over 15 rows, one (channel, tile column) per lane ------------
            if (g) __syncthreads();  // the previous group's sums have been consumed
            FFL_T(3)   // (g = 1) wait for the other waves' phase H
            const bool vlane = tid < nc * LW;
            float v[LH];
            if (vlane) {
                // wave-uniform 64-bit base (a scalar register pair) + a 32-bit per-lane byte offset: the
                // global_load saddr form -- no per-lane 64-bit address arithmetic, and the row term is one
                // 32-bit scalar multiply (a pair's 5 M planes are far below 4 GB)
                const char *Mbb = reinterpret_cast<const char *>(Mb);
                const unsigned lane_byte = ((unsigned)((c0 + vcc) * plane) + (unsigned)vgx) * 4u;
                if (FIRST) {
#pragma unroll
                    for (int j = 0; j < NCARRY; j++) v[j] = carry[g][j];
#pragma unroll
                    for (int j = 0; j < TH; j++)
                        v[NCARRY + j] = g == 0 ? sM012[(vcc * TH + j) * UP + vtx] : sM34[c0 - GC + vcc][j][vtx];
                } else {
                    if (rb == 0) {
#pragma unroll
                        for (int j = 0; j < NCARRY; j++) {
                            const unsigned gy = (unsigned)min(max(y0 + j - FFL_WIN_R, 0), h - 1);
                            v[j] = *reinterpret_cast<const float *>(Mbb + (gy * pitch + lane_byte));
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NCARRY; j++) v[j] = carry[g][j];
                    }
#pragma unroll
                    for (int j = NCARRY; j < LH; j++) {
                        const unsigned gy = (unsigned)min(max(y0 + j - FFL_WIN_R, 0), h - 1);
                        v[j] = *reinterpret_cast<const float *>(Mbb + (gy * pitch + lane_byte));
                    }
                }
            }
            if (FIRST && g == 0) __syncthreads();  // the staged rows alias the column sums written next
            if (vlane) {
#pragma unroll
                for (int j = 0; j < NCARRY; j++) carry[g][j] = v[TH + j];
                double o[TH];
                ffl_box_block16(v, o);
                const int pos = vtx + 2 * (vtx >> 4);
#pragma unroll
                for (int j = 0; j < TH; j++) sS[(vcc * TH + j) * (2 * LW2) + pos] = o[j];
            }
            FFL_T(4)   // phase V work
            __syncthreads();
            FFL_T(5)   // wait after V
            // ---- phase H: 4 window sums per lane from 18 column sums (9 x ds_read_b128) -------------
#pragma unroll
            for (int cc = 0; cc < GC; cc++) {
                if (cc >= nc) break;
                double d[PX + 14];
                const double2 *src = &sS2[cc][ty][9 * kk];
#define FFL_K5_H(Q)                                                                        \
    {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < (PX + 14) / 2; j++) {                        \
            const double2 t2 = src[2 * (Q) + j + ((4 * (Q) + 2 * j) >> 4)];                \
            d[2 * j] = t2.x;                                                               \
            d[2 * j + 1] = t2.y;                                                           \
        }                                                                                  \
        ffl_box_quarter<Q>(d, acc[c0 + cc]);                                               \
    }
                switch (q) {
                    case 0: FFL_K5_H(0) break;
                    case 1: FFL_K5_H(1) break;
                    case 2: FFL_K5_H(2) break;
                    default: FFL_K5_H(3) break;
                }
#undef FFL_K5_H
            }
        }

        FFL_T(6)   // phase H work (both groups)
        // ---- solve (+ fused UpdateMatrices) -----------------------------------------------------------
        // The solved displacements are transposed through LDS (aliasing the column-sum buffer) so that
        // the global phase below runs with lanes along x: 512-B flow rows per wave store, and the
        // bilinear gathers of R1 by neighbouring lanes fall into neighbouring addresses.
        __syncthreads();  // every lane has finished reading the column sums
        FFL_T(7)   // wait before the solve
        {
            float2 f[PX];
#pragma unroll
            for (int p = 0; p < PX; p++) {
                double g11 = acc[0][p] * scale, g12 = acc[1][p] * scale, g22 = acc[2][p] * scale,
                       h1 = acc[3][p] * scale, h2 = acc[4][p] * scale;
                double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
                f[p].x = (float)((g11 * h2 - g12 * h1) * idet);
                f[p].y = (float)((g22 * h1 - g12 * h2) * idet);
            }
            sF4[(ty * FP + lx0) / 2] = make_float4(f[0].x, f[0].y, f[1].x, f[1].y);
            sF4[(ty * FP + lx0) / 2 + 1] = make_float4(f[2].x, f[2].y, f[3].x, f[3].y);
        }
        FFL_T(8)   // solve work
        __syncthreads();
        FFL_T(9)   // wait after the solve
        if (UPDATE) {
            // One pixel per lane, lanes along x (a wave = one 64-pixel tile row, wave q takes rows q, q+4, q+8, q+12),
            // double-buffered: the R0 values and R1 corners of the lane's next row are requested before the current
            // row's arithmetic, so only the first of the four round trips is exposed.  No divergent branch touches a
            // loaded value (ffl_um_finish uses selects), so the waits are "all but the next row's 25", never "all".
            struct UmRow {
                float2 f;
                float a00, a01, a10, a11;
                bool inside;
                float r0[5];
                ffl_f2u t[5], u[5];
            };
            const int lx = tid & 63, x = min(x0 + lx, w - 1);
            const bool xin = x0 + lx < w;
            const float2 *sF2 = reinterpret_cast<const float2 *>(sF4);
            auto issue = [&](int k) {
                UmRow S;
                const int ly = (tid >> 6) + 4 * k;
                const int y = min(y0 + ly, h - 1);
                S.f = sF2[ly * FP + lx];
                const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
#pragma unroll
                for (int c = 0; c < 5; c++) S.r0[c] = *ffl_at<float>(R0 + c * plane, o);
                const UmLoc L = ffl_um_locate(w, h, x, y, S.f.x, S.f.y);
                S.a00 = L.a00; S.a01 = L.a01; S.a10 = L.a10; S.a11 = L.a11; S.inside = L.inside;
                const unsigned o1 = L.inside ? (unsigned)L.y1 * (unsigned)w + (unsigned)L.x1 : 0u;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    S.t[c] = ffl_ld_corner(R1 + c * plane, o1);
                    S.u[c] = ffl_ld_corner(R1 + c * plane, o1 + (unsigned)w);
                }
                return S;
            };
            UmRow cur = issue(0);
#pragma unroll
            for (int k = 0; k < TH / 4; k++) {
                UmRow nxt;
                if (k + 1 < TH / 4) nxt = issue(k + 1);
                const int ly = (tid >> 6) + 4 * k;
                const bool in = xin && y0 + ly < h;
                const int y = min(y0 + ly, h - 1);
                const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
                // the displacement is consumed by the UpdateMatrices right here and the next iteration reads only M: the
                // field itself is dead until the level's last iteration (8 B per pixel and launch not written; the debug
                // capture asks for it)
                if (in && store_flow) *reinterpret_cast<float2 *>(ffl_at<float>(reinterpret_cast<float *>(flow), 2u * o)) = cur.f;
                float bb[5];
#pragma unroll
                for (int c = 0; c < 5; c++) bb[c] = cur.a00 * cur.t[c].x + cur.a01 * cur.t[c].y + cur.a10 * cur.u[c].x + cur.a11 * cur.u[c].y;
                float m[5];
                ffl_um_finish(cur.r0, bb, cur.inside, w, h, x, y, cur.f.x, cur.f.y, m);
                if (in) {
#pragma unroll
                    for (int c = 0; c < 5; c++) *ffl_at<float>(Mo + c * plane, o) = m[c];
                }
                if (k + 1 < TH / 4) cur = nxt;
            }
        } else {
            // the level's last iteration: only the field is written -- two adjacent pixels per lane, 16-byte stores
            // (512-B rows per wave), the workgroup covers 8 rows per pass
            const int lx = 2 * (tid & 31), x = x0 + lx;
            if (x < w) {
                const bool second = x + 1 < w;
#pragma unroll
                for (int k = 0; k < TH / 8; k++) {
                    const int ly = (tid >> 5) + 8 * k;
                    if (y0 + ly >= h) continue;
                    const float4 ff = sF4[(ly * FP + lx) >> 1];
                    const size_t o = (size_t)(y0 + ly) * w + x;
                    if (second) {
                        ffl_f4u t;
                        t.x = ff.x; t.y = ff.y; t.z = ff.z; t.w = ff.w;
                        *reinterpret_cast<ffl_f4u *>(flow + o) = t;
                    } else {
                        flow[o] = make_float2(ff.x, ff.y);
                    }
                }
            }
        }
        FFL_T(10)  // store + fused UpdateMatrices (issue + waits for its own loads)
        __syncthreads();  // sF has been read: the next tile's column sums may overwrite it
        FFL_T(11)  // wait at the end of the tile
    }
#ifdef FFL_STAMP
    if (FFL_STAMP_ON && tid == 0)
        for (int i = 0; i < FFL_NSTAMP; i++) g_stamp[blockIdx.x & 63][i] = sStamp[i];
#endif
}

// Tiles a workgroup walks down.  The strip walk saves 14 of 30 row loads per tile after a strip's first, so long strips
// are cheaper -- as long as the launch still has enough workgroups to fill the device for several rounds.  A column of
// tiles_y tiles is cut into s equal strips (nrb = ceil(tiles_y / s)), with the smallest s that gives at least
// `min_wgs` workgroups (ffl_set_option "blur_min_wgs", default 3500).  Same-box interleaved sweep (profiles/r03_c_strip_sweep.txt,
// pairs/s against the threshold 8000, which is about what the power-of-two rule of rounds 1-2 chose): 1080p B = 32:
// 2500 -3.0 %, 3000 +1.4, 3400-3840 +1.8, 4000 +0.6; 3840x2160 B = 32: 3000-3500 +2.8, 4000 +2.4, 6000 -0.1; 256x256
// B = 256: 3000 -14.5 (a level lands on exactly 3072 workgroups), 3500-4000 +1.5, 6000 -2.2.
static int ffl_blur_rows_per_wg(int tiles_x, int tiles_y, int nB, int min_wgs) {
    // The threshold was tuned at B = 32.  On levels with many tiles per pair the strip length it gives there is also the best
    // one for larger batches (1080p level 0: strips of 17 tiles at B = 32, 48, 64, 96 -- with the batch's own count the rule
    // picks 23 / 34 tiles and the step loses 5 %, profiles/r04_strip_plain_and_batch_sweep.txt), so such levels count at most
    // 32 pairs; small levels, where a large batch is what fills the device at all (256x256, B = 256), count the batch.
    if (nB > 32 && tiles_x * tiles_y >= 256) nB = 32;
    for (int s = 1; s <= tiles_y; s++) {
        const int nrb = (tiles_y + s - 1) / s;
        if (nrb > 64) continue;
        if ((long)tiles_x * ((tiles_y + nrb - 1) / nrb) * nB >= min_wgs || nrb == 1) return nrb;
    }
    return 1;
}

void ffl_launch_blur_solve(const float *Min, float *Mout, size_t M_stride, const float *R, size_t R_stride,
                           size_t plane, const PairTab *pt, int level, int nB, int lw, int lh, int update, int store_flow,
                           const FflOptions &opt, hipStream_t st) {
    const int tiles_x = (lw + 63) / 64, tiles_y = (lh + 15) / 16;
    const int nrb = opt.blur_rows > 0 ? opt.blur_rows : ffl_blur_rows_per_wg(tiles_x, tiles_y, nB, opt.blur_min_wgs);
    dim3 grid(ffl_tile_grid(tiles_x, (tiles_y + nrb - 1) / nrb, nB));
    if (update)
        hipLaunchKernelGGL((k_blur_solve<true, 0>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride, plane, pt, level, lw,
                           lh, nrb, 0, 0, 1.0, 1.0, store_flow, nB, opt.tile_order);
    else
        hipLaunchKernelGGL((k_blur_solve<false, 0>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride, plane, pt,
                           level, lw, lh, nrb, 0, 0, 1.0, 1.0, 1, nB, opt.tile_order);
}

// first iteration of a level with the initial UpdateMatrices folded in (pw > 0: initial flow = x2 upsample of
// pt.prev, pw x ph; pw == 0: zero flow); writes the solved flow to pt.flow and the next M to Mout
void ffl_launch_blur_solve_first(float *Mout, size_t M_stride, const float *R, size_t R_stride, size_t plane,
                                 const PairTab *pt, int level, int nB, int lw, int lh, int pw, int ph, const FflOptions &opt,
                                 hipStream_t st) {
    const int tiles_x = (lw + 63) / 64, tiles_y = (lh + 15) / 16;
    const int nrb = opt.blur_rows > 0 ? opt.blur_rows : ffl_blur_rows_per_wg(tiles_x, tiles_y, nB, opt.blur_min_wgs);
    dim3 grid(ffl_tile_grid(tiles_x, (tiles_y + nrb - 1) / nrb, nB));
    if (pw > 0)
        hipLaunchKernelGGL((k_blur_solve<true, 1>), grid, dim3(256), 0, st, (const float *)nullptr, Mout, M_stride, R,
                           R_stride, plane, pt, level, lw, lh, nrb, pw, ph, (double)pw / lw, (double)ph / lh, 0, nB, opt.tile_order);
    else
        hipLaunchKernelGGL((k_blur_solve<true, 2>), grid, dim3(256), 0, st, (const float *)nullptr, Mout, M_stride, R,
                           R_stride, plane, pt, level, lw, lh, nrb, 0, 0, 1.0, 1.0, 0, nB, opt.tile_order);
}
