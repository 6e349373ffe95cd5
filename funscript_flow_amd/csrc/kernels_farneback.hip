// gfx950 kernels for dense Farneback flow (what cv2.calcOpticalFlowFarneback computes at
// FunscriptFlow.pyw:878-879).  Built with -ffp-contract=off: every float/double operation below is
// meant literally, in the order written, so that results are bit-identical to the CPU oracle.
//
// Kernel            roofline   algorithmic bytes / level pixel
//   k_gray          HBM        4 per full-res pixel (3 in, 1 out)
//   k_pyr_level     HBM/L1     1 per full-res pixel in + 4 per level pixel out
//   k_polyexp       HBM        24  (4 in, 20 out), 11x11 separable through LDS
//   k_flow_upsample HBM        10  (2 in at quarter res, 8 out)
//   k_update_mat    HBM        68  (R0 20 + R1 gather 20 + flow 8 -> M 20)
//   k_blur_solve    HBM        28  (M 20 -> flow 8) [+68 when the next UpdateMatrices is fused]
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

__device__ __forceinline__ void ffl_resize_coord(int d, int src, int dst, int &i0, int &i1, float &f) {
    double scale = (double)src / dst;
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
    i0 = sx;
    i1 = sx + 1 < src ? sx + 1 : src - 1;
    f = fx;
}

__device__ __forceinline__ float ffl_blur_at(const uint8_t *__restrict__ img, int w, int h, int sx, int sy,
                                             const GaussKernel &gk) {
    const int r = gk.ksize >> 1;
    float acc = 0.f;
    // vertical combination of horizontally blurred rows, in the oracle's order: centre row first,
    // then the symmetric pairs j = 1..r
    for (int j = 0; j <= r; j++) {
        float hv[2];
        const int rows[2] = {ffl_reflect101(sy - j, h), ffl_reflect101(sy + j, h)};
        const int nrow = j == 0 ? 1 : 2;
        for (int q = 0; q < nrow; q++) {
            const uint8_t *p = img + (size_t)rows[q] * w;
            float a = gk.k[r] * (float)p[sx];
            for (int i = 1; i <= r; i++) {
                float lo = (float)p[ffl_reflect101(sx - i, w)];
                float hi = (float)p[ffl_reflect101(sx + i, w)];
                a = a + gk.k[r + i] * (lo + hi);
            }
            hv[q] = a;
        }
        if (j == 0) acc = gk.k[r] * hv[0];
        else acc = acc + gk.k[r + j] * (hv[0] + hv[1]);
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_pyr_level(const uint8_t *__restrict__ gray_base, size_t gray_stride, UTab ut,
                                                   int w, int h, int lw, int lh, GaussKernel gk,
                                                   float *__restrict__ I, size_t I_stride) {
    int dx = blockIdx.x * 32 + (threadIdx.x & 31);
    int dy = blockIdx.y * 8 + (threadIdx.x >> 5);
    int u = blockIdx.z;
    if (dx >= lw || dy >= lh) return;
    const uint8_t *img = gray_base + (size_t)ut.fslot[u] * gray_stride;
    int x0, x1, y0, y1;
    float a1, b1;
    ffl_resize_coord(dx, w, lw, x0, x1, a1);
    ffl_resize_coord(dy, h, lh, y0, y1, b1);
    float a0 = 1.f - a1, b0 = 1.f - b1;
    // samples with weight exactly 0 are skipped: v*1 + s*0 == v for the finite non-negative s here
    float v00 = ffl_blur_at(img, w, h, x0, y0, gk);
    float v01 = a1 != 0.f ? ffl_blur_at(img, w, h, x1, y0, gk) : 0.f;
    float t0 = v00 * a0 + v01 * a1;
    float t1 = 0.f;
    if (b1 != 0.f) {
        float v10 = ffl_blur_at(img, w, h, x0, y1, gk);
        float v11 = a1 != 0.f ? ffl_blur_at(img, w, h, x1, y1, gk) : 0.f;
        t1 = v10 * a0 + v11 * a1;
    }
    I[(size_t)u * I_stride + (size_t)dy * lw + dx] = t0 * b0 + t1 * b1;
}

void ffl_launch_pyr_level(const uint8_t *gray_base, size_t gray_stride, UTab ut, int nU, int w, int h, int lw, int lh,
                          GaussKernel gk, float *I, size_t I_stride, hipStream_t st) {
    dim3 grid((lw + 31) / 32, (lh + 7) / 8, nU);
    hipLaunchKernelGGL(k_pyr_level, grid, dim3(256), 0, st, gray_base, gray_stride, ut, w, h, lw, lh, gk, I, I_stride);
}

// ------------------------------------------------------------------------------------------------
// K2: polynomial expansion (FarnebackPolyExp, n = 5, sigma = 1.2): I (lh,lw) -> R 5 planes.
// One 64x16 output tile per 256-thread workgroup; the (64+10)x(16+10) input tile and the three
// vertically filtered rows live in LDS; the horizontal pass accumulates in double.
// ------------------------------------------------------------------------------------------------
#define PE_TW 64
#define PE_TH 16
#define PE_N FFL_POLY_N
#define PE_LW (PE_TW + 2 * PE_N)

__global__ __launch_bounds__(256) void k_polyexp(const float *__restrict__ I, size_t I_stride, float *__restrict__ R,
                                                 size_t R_stride, size_t plane, int w, int h, PolyConsts pc) {
    __shared__ float sI[PE_TH + 2 * PE_N][PE_LW];
    __shared__ float sV[3][PE_TH][PE_LW + 1];
    const int tid = threadIdx.x;
    const int u = blockIdx.z;
    const int x0 = blockIdx.x * PE_TW, y0 = blockIdx.y * PE_TH;
    const float *img = I + (size_t)u * I_stride;

    for (int i = tid; i < (PE_TH + 2 * PE_N) * PE_LW; i += 256) {
        int ly = i / PE_LW, lx = i - ly * PE_LW;
        int gy = min(max(y0 + ly - PE_N, 0), h - 1), gx = min(max(x0 + lx - PE_N, 0), w - 1);
        sI[ly][lx] = img[(size_t)gy * w + gx];
    }
    __syncthreads();

    // vertical part (float): rows are clamped because the tile was loaded with clamped rows
    for (int i = tid; i < PE_TH * PE_LW; i += 256) {
        int ly = i / PE_LW, lx = i - ly * PE_LW;
        float c = sI[ly + PE_N][lx];
        float r0 = c * pc.g[0], r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int k = 1; k <= PE_N; k++) {
            float a = sI[ly + PE_N - k][lx], b = sI[ly + PE_N + k][lx];
            float p = a + b;
            r0 = r0 + pc.g[k] * p;
            r1 = r1 + pc.xg[k] * (b - a);
            r2 = r2 + pc.xxg[k] * p;
        }
        sV[0][ly][lx] = r0;
        sV[1][ly][lx] = r1;
        sV[2][ly][lx] = r2;
    }
    __syncthreads();

    // horizontal part (double accumulators; the b2,b3,b5,b6 products are float products)
    float *out = R + (size_t)u * R_stride;
    for (int i = tid; i < PE_TH * PE_TW; i += 256) {
        int ly = i / PE_TW, lx = i - ly * PE_TW;
        int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) continue;
        const float *v0 = &sV[0][ly][lx + PE_N], *v1 = &sV[1][ly][lx + PE_N], *v2 = &sV[2][ly][lx + PE_N];
        float g0 = pc.g[0];
        double b1 = (double)(v0[0] * g0), b2 = 0, b3 = (double)(v1[0] * g0), b4 = 0, b5 = (double)(v2[0] * g0), b6 = 0;
#pragma unroll
        for (int k = 1; k <= PE_N; k++) {
            double tg = (double)(v0[k] + v0[-k]);
            g0 = pc.g[k];
            b1 += tg * (double)g0;
            b4 += tg * (double)pc.xxg[k];
            b2 += (double)((v0[k] - v0[-k]) * pc.xg[k]);
            b3 += (double)((v1[k] + v1[-k]) * g0);
            b6 += (double)((v1[k] - v1[-k]) * pc.xg[k]);
            b5 += (double)((v2[k] + v2[-k]) * g0);
        }
        size_t o = (size_t)y * w + x;
        out[o] = (float)(b3 * pc.ig11);
        out[plane + o] = (float)(b2 * pc.ig11);
        out[2 * plane + o] = (float)(b1 * pc.ig03 + b5 * pc.ig33);
        out[3 * plane + o] = (float)(b1 * pc.ig03 + b4 * pc.ig33);
        out[4 * plane + o] = (float)(b6 * pc.ig55);
    }
}

void ffl_launch_polyexp(const float *I, size_t I_stride, float *R, size_t R_stride, size_t plane, int nU, int lw,
                        int lh, PolyConsts pc, hipStream_t st) {
    dim3 grid((lw + PE_TW - 1) / PE_TW, (lh + PE_TH - 1) / PE_TH, nU);
    hipLaunchKernelGGL(k_polyexp, grid, dim3(256), 0, st, I, I_stride, R, R_stride, plane, lw, lh, pc);
}

// ------------------------------------------------------------------------------------------------
// K3: flow = resize(prevFlow, (lw, lh), INTER_LINEAR) * (1 / pyrScale)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_upsample(PairTab pt, int pw, int ph, int lw, int lh) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int b = blockIdx.z;
    if (x >= lw || y >= lh) return;
    const float2 *prev = reinterpret_cast<const float2 *>(pt.prev[b]);
    float2 *flow = reinterpret_cast<float2 *>(pt.flow[b]);
    int x0, x1, y0, y1;
    float a1, b1;
    ffl_resize_coord(x, pw, lw, x0, x1, a1);
    ffl_resize_coord(y, ph, lh, y0, y1, b1);
    float a0 = 1.f - a1, b0 = 1.f - b1;
    float2 p00 = prev[(size_t)y0 * pw + x0], p01 = prev[(size_t)y0 * pw + x1];
    float2 p10 = prev[(size_t)y1 * pw + x0], p11 = prev[(size_t)y1 * pw + x1];
    float2 o;
    {
        float t0 = p00.x * a0 + p01.x * a1, t1 = p10.x * a0 + p11.x * a1;
        o.x = (t0 * b0 + t1 * b1) * 2.0f;
    }
    {
        float t0 = p00.y * a0 + p01.y * a1, t1 = p10.y * a0 + p11.y * a1;
        o.y = (t0 * b0 + t1 * b1) * 2.0f;
    }
    flow[(size_t)y * lw + x] = o;
}

void ffl_launch_flow_upsample(PairTab pt, int nB, int pw, int ph, int lw, int lh, hipStream_t st) {
    dim3 grid((lw + 63) / 64, (lh + 3) / 4, nB);
    hipLaunchKernelGGL(k_flow_upsample, grid, dim3(256), 0, st, pt, pw, ph, lw, lh);
}

// ------------------------------------------------------------------------------------------------
// K4: FarnebackUpdateMatrices (standalone; runs once per level before the first blur iteration)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_matrices(const float *__restrict__ R, size_t R_stride, size_t plane,
                                                         PairTab pt, float *__restrict__ M, size_t M_stride, int w,
                                                         int h) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int b = blockIdx.z;
    if (x >= w || y >= h) return;
    const float *R0 = R + (size_t)pt.u0[b] * R_stride, *R1 = R + (size_t)pt.u1[b] * R_stride;
    float2 f = reinterpret_cast<const float2 *>(pt.flow[b])[(size_t)y * w + x];
    float m[5];
    ffl_um_pixel(R0, R1, plane, w, h, x, y, f.x, f.y, m);
    float *Mo = M + (size_t)b * M_stride + (size_t)y * w + x;
#pragma unroll
    for (int c = 0; c < 5; c++) Mo[c * plane] = m[c];
}

void ffl_launch_update_matrices(const float *R, size_t R_stride, size_t plane, PairTab pt, int nB, float *M,
                                size_t M_stride, int lw, int lh, hipStream_t st) {
    dim3 grid((lw + 63) / 64, (lh + 3) / 4, nB);
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, st, R, R_stride, plane, pt, M, M_stride, lw, lh);
}

// ------------------------------------------------------------------------------------------------
// K5: FarnebackUpdateFlow_Blur: 15x15 box sum of the 5 M planes (double, REPLICATE border, fixed
// order rows then columns), 2x2 solve in double, flow write, and -- when UPDATE -- the next
// UpdateMatrices fused on the freshly solved displacement (written to the other M buffer, so
// neighbouring tiles still read the old M: a Jacobi step, identical to OpenCV's striped update).
//
// Per channel: the (TH+14)x(TW+14) float tile is staged in LDS, column sums over 15 rows go to an
// LDS double buffer, and each lane then adds 15 neighbouring column sums for its pixels.
// ------------------------------------------------------------------------------------------------
template <int TW, int TH, bool UPDATE>
__global__ __launch_bounds__(256) void k_blur_solve(const float *__restrict__ Min, float *__restrict__ Mout,
                                                    size_t M_stride, const float *__restrict__ R, size_t R_stride,
                                                    size_t plane, PairTab pt, int w, int h) {
    constexpr int LW = TW + 2 * FFL_WIN_R, LH = TH + 2 * FFL_WIN_R;
    constexpr int PPT = TW * TH / 256;  // pixels per thread
    static_assert(TW == 64 && (TW * TH) % 256 == 0, "tile shape");
    __shared__ float sT[LH][LW];
    __shared__ double sS[TH][LW];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const float *Mb = Min + (size_t)b * M_stride;
    const int lx = tid & 63, lyb = tid >> 6;  // pixel p of this thread: (lx, lyb + 4*p)

    double acc[5][PPT];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float *Mc = Mb + (size_t)c * plane;
#pragma unroll 1
        for (int i = tid; i < LH * LW; i += 256) {
            int ty = i / LW, tx = i - ty * LW;
            int gy = min(max(y0 + ty - FFL_WIN_R, 0), h - 1), gx = min(max(x0 + tx - FFL_WIN_R, 0), w - 1);
            sT[ty][tx] = Mc[(size_t)gy * w + gx];
        }
        __syncthreads();
#pragma unroll 1
        for (int i = tid; i < TH * LW; i += 256) {
            int ty = i / LW, tx = i - ty * LW;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < FFL_WIN; j++) s += (double)sT[ty + j][tx];
            sS[ty][tx] = s;
        }
        __syncthreads();
        {
            // PPT independent accumulators, each summed in column order j = 0..14; the j loop is kept
            // rolled so that only PPT LDS reads are in flight (fully unrolled it needs 60 doubles)
            double s[PPT];
#pragma unroll
            for (int p = 0; p < PPT; p++) s[p] = 0.0;
#pragma unroll 1
            for (int j = 0; j < FFL_WIN; j++) {
#pragma unroll
                for (int p = 0; p < PPT; p++) s[p] += sS[lyb + 4 * p][lx + j];
            }
#pragma unroll
            for (int p = 0; p < PPT; p++) acc[c][p] = s[p];
        }
        __syncthreads();
    }

    const double scale = 1.0 / (FFL_WIN * FFL_WIN);
    float2 *flow = reinterpret_cast<float2 *>(pt.flow[b]);
    const float *R0 = R + (size_t)pt.u0[b] * R_stride, *R1 = R + (size_t)pt.u1[b] * R_stride;
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        int x = x0 + lx, y = y0 + lyb + 4 * p;
        if (x >= w || y >= h) continue;
        double g11 = acc[0][p] * scale, g12 = acc[1][p] * scale, g22 = acc[2][p] * scale, h1 = acc[3][p] * scale,
               h2 = acc[4][p] * scale;
        double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
        float2 f;
        f.x = (float)((g11 * h2 - g12 * h1) * idet);
        f.y = (float)((g22 * h1 - g12 * h2) * idet);
        flow[(size_t)y * w + x] = f;
        if (UPDATE) {
            float m[5];
            ffl_um_pixel(R0, R1, plane, w, h, x, y, f.x, f.y, m);
            float *Mo = Mout + (size_t)b * M_stride + (size_t)y * w + x;
#pragma unroll
            for (int c = 0; c < 5; c++) Mo[c * plane] = m[c];
        }
    }
}

void ffl_launch_blur_solve(const float *Min, float *Mout, size_t M_stride, const float *R, size_t R_stride,
                           size_t plane, PairTab pt, int nB, int lw, int lh, int update, hipStream_t st) {
    constexpr int TW = 64, TH = 16;
    dim3 grid((lw + TW - 1) / TW, (lh + TH - 1) / TH, nB);
    if (update)
        hipLaunchKernelGGL((k_blur_solve<TW, TH, true>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride, plane,
                           pt, lw, lh);
    else
        hipLaunchKernelGGL((k_blur_solve<TW, TH, false>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride,
                           plane, pt, lw, lh);
}
