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

// `scale` = (double)src / dst, formed once on the host (IEEE division: the same double the oracle forms)
__device__ __forceinline__ void ffl_resize_coord(int d, int src, double scale, int &i0, int &i1, float &f) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
    i0 = sx;
    i1 = sx + 1 < src ? sx + 1 : src - 1;
    f = fx;
}

// One OW x OH output tile per 256-thread workgroup, three LDS stages:
//   stage   the full-resolution source window the tile samples (+ blur radius), uint8 -> float,
//           REFLECT_101 resolved here so the filter loops below are branch-free;
//   H pass  horizontal blur of every staged row at the (<= 2 per output) sampled columns; lanes run
//           along rows and the row pitch is odd, so the strided column reads are conflict-free;
//   V pass  vertical blur at the (<= 2) sampled rows + the two lerps, one output per lane.
// Samples whose lerp weight is exactly 0 are not evaluated: v*1 + s*0 == v for finite s >= 0.
struct PyrTile {
    int OW, OH;   // output tile (OW a power of two)
    int SW, SH;   // staged source window (max over tiles), SW odd
    int lgOW;
    double sx, sy;  // (double)w / lw, (double)h / lh
};

// R > 0: blur radius known at compile time (taps unrolled, coefficients read once from the kernel
// arguments into scalar registers); R == 0: runtime radius.
template <int R>
__global__ __launch_bounds__(256) void k_pyr_level(const uint8_t *__restrict__ gray_base, size_t gray_stride, UTab ut,
                                                   int w, int h, int lw, int lh, GaussKernel gk,
                                                   float *__restrict__ I, size_t I_stride, PyrTile pt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int OW = pt.OW, OH = pt.OH, SW = pt.SW, SH = pt.SH;
    const int HP = SH | 1;                      // row pitch (in columns) of the H-pass planes: odd
    float *sSrc = smem;                         // [SH][SW]
    float *sH = sSrc + SH * SW;                 // [2][OW][HP]  (q, output column, staged row)
    int *sX0 = reinterpret_cast<int *>(sH + 2 * OW * HP);  // per output column: x0, x1
    int *sX1 = sX0 + OW;
    float *sFX = reinterpret_cast<float *>(sX1 + OW);
    int *sY0 = reinterpret_cast<int *>(sFX + OW);
    int *sY1 = sY0 + OH;
    float *sFY = reinterpret_cast<float *>(sY1 + OH);

    const int tid = threadIdx.x, u = blockIdx.z;
    const int dx0 = blockIdx.x * OW, dy0 = blockIdx.y * OH;
    const int nx = min(OW, lw - dx0), ny = min(OH, lh - dy0);
    const int r = R > 0 ? R : (gk.ksize >> 1);
    const uint8_t *img = gray_base + (size_t)ut.fslot[u] * gray_stride;

    if (tid < nx) {
        int a, b;
        float f;
        ffl_resize_coord(dx0 + tid, w, pt.sx, a, b, f);
        sX0[tid] = a; sX1[tid] = b; sFX[tid] = f;
    }
    if (tid >= 64 && tid - 64 < ny) {
        int a, b;
        float f;
        ffl_resize_coord(dy0 + tid - 64, h, pt.sy, a, b, f);
        sY0[tid - 64] = a; sY1[tid - 64] = b; sFY[tid - 64] = f;
    }
    __syncthreads();
    const int xs = sX0[0] - r, ys = sY0[0] - r;
    const int span_w = sX1[nx - 1] + r - xs + 1, span_h = sY1[ny - 1] + r - ys + 1;  // <= SW, SH by construction

    const int lane = tid & 63, wv = tid >> 6;  // all loops below are (wave, lane) nests: no divisions
    for (int j = wv; j < span_h; j += 4) {
        const uint8_t *row = img + (size_t)ffl_reflect101(ys + j, h) * w;
        for (int c = lane; c < span_w; c += 64) sSrc[j * SW + c] = (float)row[ffl_reflect101(xs + c, w)];
    }
    __syncthreads();

    // H pass -> sH[q][d][j].  q = 1 (the right lerp neighbour) is needed only when resampling in x.
    const int nq = (lw != w) ? 2 : 1;
    if (pt.OW >= 32) {
        // mild decimation: lanes along output columns (source stride <= 4 floats), waves along rows
        for (int q = 0; q < nq; q++)
            for (int d = lane; d < nx; d += 64) {
                const bool on = q == 0 || sFX[d] != 0.f;
                const float *p0 = sSrc + ((q ? sX1[d] : sX0[d]) - xs);
                for (int j = wv; j < span_h; j += 4) {
                    float acc = 0.f;
                    if (on) {
                        const float *p = p0 + j * SW;
                        acc = gk.k[r] * p[0];
#pragma unroll
                        for (int t = 1; t <= r; t++) acc = acc + gk.k[r + t] * (p[-t] + p[t]);
                    }
                    sH[(q * OW + d) * HP + j] = acc;
                }
            }
    } else {
        // strong decimation: lanes along staged rows (odd pitch: conflict-free), waves along (q, d)
        for (int dq = wv; dq < nq * nx; dq += 4) {
            const int q = dq >= nx, d = q ? dq - nx : dq;
            const bool on = q == 0 || sFX[d] != 0.f;
            const float *p0 = sSrc + ((q ? sX1[d] : sX0[d]) - xs);
            for (int j = lane; j < span_h; j += 64) {
                float acc = 0.f;
                if (on) {
                    const float *p = p0 + j * SW;
                    acc = gk.k[r] * p[0];
#pragma unroll
                    for (int t = 1; t <= r; t++) acc = acc + gk.k[r + t] * (p[-t] + p[t]);
                }
                sH[(q * OW + d) * HP + j] = acc;
            }
        }
    }
    __syncthreads();

    // V pass + lerps: one output per lane, output columns fastest (OW is a power of two)
    for (int i = tid; i < OW * OH; i += 256) {
        const int oy = i >> pt.lgOW, ox = i & (OW - 1);
        if (ox >= nx || oy >= ny) continue;
        float a1 = sFX[ox], b1 = sFY[oy], a0 = 1.f - a1, b0 = 1.f - b1;
        const float *h0 = sH + (0 * OW + ox) * HP, *h1 = sH + (1 * OW + ox) * HP;
        float t[2] = {0.f, 0.f};
#pragma unroll
        for (int qy = 0; qy < 2; qy++) {
            if (qy == 1 && b1 == 0.f) break;
            int cy = (qy ? sY1[oy] : sY0[oy]) - ys;
            float v0 = gk.k[r] * h0[cy];
#pragma unroll
            for (int j = 1; j <= r; j++) v0 = v0 + gk.k[r + j] * (h0[cy - j] + h0[cy + j]);
            float v1 = 0.f;
            if (a1 != 0.f) {
                v1 = gk.k[r] * h1[cy];
#pragma unroll
                for (int j = 1; j <= r; j++) v1 = v1 + gk.k[r + j] * (h1[cy - j] + h1[cy + j]);
            }
            t[qy] = v0 * a0 + v1 * a1;
        }
        I[(size_t)u * I_stride + (size_t)(dy0 + oy) * lw + dx0 + ox] = t[0] * b0 + t[1] * b1;
    }
}

// host copy of the device coordinate rule, used to size the staged window exactly
static void host_resize_coord(int d, int src, int dst, int &i0, int &i1) {
    double scale = (double)src / dst;
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    if (sx < 0) sx = 0;
    if (sx >= src - 1) sx = src - 1;
    i0 = sx;
    i1 = sx + 1 < src ? sx + 1 : src - 1;
}

static int max_span(int src, int dst, int tile, int r) {
    int best = 0;
    for (int d0 = 0; d0 < dst; d0 += tile) {
        int a0, a1, b0, b1;
        host_resize_coord(d0, src, dst, a0, a1);
        host_resize_coord(d0 + tile < dst ? d0 + tile - 1 : dst - 1, src, dst, b0, b1);
        int span = b1 - a0 + 1 + 2 * r;
        if (span > best) best = span;
    }
    return best;
}

void ffl_launch_pyr_level(const uint8_t *gray_base, size_t gray_stride, UTab ut, int nU, int w, int h, int lw, int lh,
                          GaussKernel gk, float *I, size_t I_stride, hipStream_t st) {
    const int r = gk.ksize / 2;
    // output tile: 64 wide at (near) full resolution, narrower as the decimation factor grows so the
    // staged window stays within ~56 KB of LDS
    PyrTile pt;
    const double sx = (double)w / lw;
    pt.OW = sx <= 2.5 ? 64 : (sx <= 5 ? 32 : 16);
    pt.OH = sx <= 1.5 ? 16 : 8;
    for (;;) {
        pt.SW = max_span(w, lw, pt.OW, r) | 1;
        pt.SH = max_span(h, lh, pt.OH, r);
        size_t bytes = sizeof(float) * ((size_t)pt.SH * pt.SW + 2 * (size_t)pt.OW * (pt.SH | 1)) +
                       sizeof(int) * 3 * (size_t)(pt.OW + pt.OH);
        if (bytes <= 60 * 1024 || (pt.OW <= 8 && pt.OH <= 4)) {
            pt.sx = (double)w / lw;
            pt.sy = (double)h / lh;
            pt.lgOW = 0;
            while ((1 << pt.lgOW) < pt.OW) pt.lgOW++;
            dim3 grid((lw + pt.OW - 1) / pt.OW, (lh + pt.OH - 1) / pt.OH, nU);
#define FFL_PYR_LAUNCH(RR)                                                                                      \
    hipLaunchKernelGGL(k_pyr_level<RR>, grid, dim3(256), bytes, st, gray_base, gray_stride, ut, w, h, lw, lh, gk, I, \
                       I_stride, pt)
            if (r == 1) FFL_PYR_LAUNCH(1);
            else if (r == 4) FFL_PYR_LAUNCH(4);
            else if (r == 9) FFL_PYR_LAUNCH(9);
            else FFL_PYR_LAUNCH(0);
#undef FFL_PYR_LAUNCH
            return;
        }
        if (pt.OW > 8) pt.OW >>= 1;
        else pt.OH >>= 1;
    }
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
// K3 + K4: FarnebackUpdateMatrices (standalone; runs once per level before the first blur
// iteration).  With UPSAMPLE the level's initial flow -- K3: resize(prevFlow, (w, h), INTER_LINEAR)
// * (1 / pyrScale) -- is produced here as well, written once and used from registers.  64x16 tiles
// in XCD-aware panel order: the R1 rows y1, y1+1 gathered by vertically adjacent tiles are re-read
// from L2.  usx / usy = (double)pw / w, (double)ph / h.
// ------------------------------------------------------------------------------------------------
template <bool UPSAMPLE>
__global__ __launch_bounds__(256) void k_update_matrices(const float *__restrict__ R, size_t R_stride, size_t plane,
                                                         PairTab pt, float *__restrict__ M, size_t M_stride, int w,
                                                         int h, int pw, int ph, double usx, double usy) {
    int b, tile_x, tile_y;
    if (!ffl_tile_coord((w + 63) / 64, (h + 15) / 16, b, tile_x, tile_y)) return;
    const int x = tile_x * 64 + (threadIdx.x & 63);
    if (x >= w) return;
    const float *R0 = R + (size_t)pt.u0[b] * R_stride, *R1 = R + (size_t)pt.u1[b] * R_stride;
    float2 *flow = reinterpret_cast<float2 *>(pt.flow[b]);
    const float2 *prev = reinterpret_cast<const float2 *>(pt.prev[b]);
    int x0 = 0, x1 = 0;
    float a1 = 0.f;
    if (UPSAMPLE) ffl_resize_coord(x, pw, usx, x0, x1, a1);
    // 4 rows per lane, branch-free (rows past the image are clamped for the loads and only their
    // stores are predicated) so that the loads of all 4 rows are in flight together.
    float2 f[4];
    int ys[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ys[k] = tile_y * 16 + (threadIdx.x >> 6) + 4 * k;
        const int y = min(ys[k], h - 1);
        const size_t o = (size_t)y * w + x;
        if (UPSAMPLE) {
            int y0, y1;
            float b1;
            ffl_resize_coord(y, ph, usy, y0, y1, b1);
            const float a0 = 1.f - a1, b0 = 1.f - b1;
            const float2 p00 = prev[(size_t)y0 * pw + x0], p01 = prev[(size_t)y0 * pw + x1];
            const float2 p10 = prev[(size_t)y1 * pw + x0], p11 = prev[(size_t)y1 * pw + x1];
            {
                float t0 = p00.x * a0 + p01.x * a1, t1 = p10.x * a0 + p11.x * a1;
                f[k].x = (t0 * b0 + t1 * b1) * 2.0f;
            }
            {
                float t0 = p00.y * a0 + p01.y * a1, t1 = p10.y * a0 + p11.y * a1;
                f[k].y = (t0 * b0 + t1 * b1) * 2.0f;
            }
            if (ys[k] < h) flow[o] = f[k];
        } else {
            f[k] = flow[o];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = min(ys[k], h - 1);
        float m[5];
        ffl_um_pixel(R0, R1, plane, w, h, x, y, f[k].x, f[k].y, m);
        if (ys[k] < h) {
            float *Mo = M + (size_t)b * M_stride + (size_t)y * w + x;
#pragma unroll
            for (int c = 0; c < 5; c++) Mo[c * plane] = m[c];
        }
    }
}

void ffl_launch_update_matrices(const float *R, size_t R_stride, size_t plane, PairTab pt, int nB, float *M,
                                size_t M_stride, int lw, int lh, int pw, int ph, hipStream_t st) {
    dim3 grid(ffl_tile_grid((lw + 63) / 64, (lh + 15) / 16, nB));
    if (pw > 0)
        hipLaunchKernelGGL(k_update_matrices<true>, grid, dim3(256), 0, st, R, R_stride, plane, pt, M, M_stride, lw, lh,
                           pw, ph, (double)pw / lw, (double)ph / lh);
    else
        hipLaunchKernelGGL(k_update_matrices<false>, grid, dim3(256), 0, st, R, R_stride, plane, pt, M, M_stride, lw, lh,
                           0, 0, 1.0, 1.0);
}

// ------------------------------------------------------------------------------------------------
// K5: FarnebackUpdateFlow_Blur: 15x15 box sum of the 5 M planes (double, REPLICATE border, rows
// then columns, each 15-term sum in the fixed pairwise tree of the oracle's box15()), 2x2 solve in
// double, flow write, and -- when UPDATE -- the next UpdateMatrices fused on the freshly solved
// displacement (written to the other M buffer, so neighbouring tiles still read the old M: a Jacobi
// step, identical to OpenCV's striped in-place update).
//
// One 64 x TH output tile per 256-thread workgroup:
//   phase V  390 lanes = 5 channels x 78 tile columns; each lane reads its TH+14 rows straight from
//            global memory (consecutive lanes = consecutive x: coalesced), forms the TH column sums
//            with shared s2/s4/s8 partials (8.6 instead of 14 additions per output) and writes them
//            to LDS as doubles;
//   phase H  each lane owns 4 consecutive pixels of one row: 9 ds_read_b128 per channel bring the
//            18 column sums it needs, the same tree gives the 4 window sums;
//   solve    2x2 system in double, float2 flow store (2 x dwordx4 per lane), fused UpdateMatrices
//            with dwordx4 R0 loads / M stores.
// ------------------------------------------------------------------------------------------------
#ifndef FFL_K5_WAVES
#define FFL_K5_WAVES 4  // waves per SIMD the register allocator must leave room for (= workgroups per CU)
#endif

template <int NOUT, typename T>
__device__ __forceinline__ void ffl_box15_run(const T (&v)[NOUT + 14], double (&out)[NOUT]) {
    double s2[NOUT + 12], s4[NOUT + 8], s8[NOUT];
    double dprev = 0.0;
#pragma unroll
    for (int t = 0; t < NOUT + 14; t++) {
        const double d = (double)v[t];  // widened at first use: only a sliding window of partials is live
        if (t >= 1 && t - 1 < NOUT + 12) s2[t - 1] = dprev + d;
        if (t >= 3 && t - 3 < NOUT + 8) s4[t - 3] = s2[t - 3] + s2[t - 1];
        if (t >= 7 && t - 7 < NOUT) s8[t - 7] = s4[t - 7] + s4[t - 3];
        if (t >= 14) out[t - 14] = ((s8[t - 14] + s4[t - 6]) + s2[t - 2]) + d;
        dprev = d;
    }
}

template <int TH, bool UPDATE>
__global__ __launch_bounds__(256, FFL_K5_WAVES) void k_blur_solve(const float *__restrict__ Min, float *__restrict__ Mout,
                                                    size_t M_stride, const float *__restrict__ R, size_t R_stride,
                                                    size_t plane, PairTab pt, int w, int h) {
    constexpr int TW = 64, LW = TW + 2 * FFL_WIN_R, LH = TH + 2 * FFL_WIN_R;
    constexpr int PX = 4;  // consecutive pixels per lane in phase H
    static_assert(TH == 8 || TH == 16, "one phase-H pass of 256 lanes covers 64 x 16 pixels");
    // The 5 channels go through LDS in two groups (3 + 2): 3*TH*78 doubles = 30 KB at TH = 16, so
    // five workgroups fit a CU, and each group's 3*78 / 2*78 column lanes fit one pass of 256 lanes.
    // double2-typed so that phase H reads with ds_read_b128 (row pitch 624 B = 39 x 16 B).
    __shared__ double2 sS2[3][TH][LW / 2];
    const int tid = threadIdx.x;
    int b, tile_x, tile_y;  // XCD-aware panel order, see ffl_tile_coord
    if (!ffl_tile_coord((w + TW - 1) / TW, (h + TH - 1) / TH, b, tile_x, tile_y)) return;
    const int x0 = tile_x * TW, y0 = tile_y * TH;
    const float *Mb = Min + (size_t)b * M_stride;

    const int xg = tid & (TW / PX - 1), ty = tid / (TW / PX);  // phase H: 4 pixels (x0+4xg.., y0+ty)
    double acc[5][PX];
    double *sS = reinterpret_cast<double *>(&sS2[0][0][0]);
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int c0 = g * 3, nc = g == 0 ? 3 : 2;
        // ---- phase V: column sums over 15 rows, one (channel, tile column) per lane ------------
        if (g) __syncthreads();  // the previous group's sums have been consumed
        if (tid < nc * LW) {
            const int cc = tid / LW, tx = tid - cc * LW;
            const int gx = min(max(x0 + tx - FFL_WIN_R, 0), w - 1);
            // wave-uniform row base (scalar registers) + one 32-bit per-lane offset for all 30 loads
            const unsigned lane_off = (unsigned)((c0 + cc) * plane) + (unsigned)gx;
            float v[LH];
#pragma unroll
            for (int j = 0; j < LH; j++) {
                const int gy = min(max(y0 + j - FFL_WIN_R, 0), h - 1);
                const float *row = Mb + (size_t)gy * w;
                v[j] = row[lane_off];
            }
            double o[TH];
            ffl_box15_run<TH>(v, o);
#pragma unroll
            for (int j = 0; j < TH; j++) sS[(cc * TH + j) * LW + tx] = o[j];
        }
        __syncthreads();
        // ---- phase H: 4 window sums per lane from 18 column sums (9 x ds_read_b128) -------------
        if (ty < TH) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                if (cc >= nc) break;
                double d[PX + 14];
                const double2 *src = &sS2[cc][ty][xg * (PX / 2)];
#pragma unroll
                for (int j = 0; j < (PX + 14) / 2; j++) {
                    double2 q = src[j];
                    d[2 * j] = q.x;
                    d[2 * j + 1] = q.y;
                }
                ffl_box15_run<PX>(d, acc[c0 + cc]);
            }
        }
    }

    // ---- solve (+ fused UpdateMatrices) ---------------------------------------------------------------
    const double scale = 1.0 / (FFL_WIN * FFL_WIN);
    float2 *flow = reinterpret_cast<float2 *>(pt.flow[b]);
    const float *R0 = R + (size_t)pt.u0[b] * R_stride, *R1 = R + (size_t)pt.u1[b] * R_stride;
    // The solved displacements are transposed through LDS (aliasing the column-sum buffer) so that
    // the global phase below runs with lanes along x: 512-B flow rows per wave store, and the
    // bilinear gathers of R1 by neighbouring lanes fall into neighbouring addresses.
    __syncthreads();  // every lane has finished reading the column sums
    float4 *sF4 = reinterpret_cast<float4 *>(&sS2[0][0][0]);  // float2 sF[TH][TW] viewed as float4 pairs
    if (ty < TH) {
        float2 f[PX];
#pragma unroll
        for (int p = 0; p < PX; p++) {
            double g11 = acc[0][p] * scale, g12 = acc[1][p] * scale, g22 = acc[2][p] * scale, h1 = acc[3][p] * scale,
                   h2 = acc[4][p] * scale;
            double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
            f[p].x = (float)((g11 * h2 - g12 * h1) * idet);
            f[p].y = (float)((g22 * h1 - g12 * h2) * idet);
        }
        sF4[(ty * TW + xg * PX) / 2] = make_float4(f[0].x, f[0].y, f[1].x, f[1].y);
        sF4[(ty * TW + xg * PX) / 2 + 1] = make_float4(f[2].x, f[2].y, f[3].x, f[3].y);
    }
    __syncthreads();
    const float2 *sF = reinterpret_cast<const float2 *>(sF4);
    const int lx = tid & 63, x = x0 + lx;
    if (x >= w) return;
    // branch-free over the lane's TH/4 rows (clamped loads, predicated stores): all gathers in flight
#pragma unroll
    for (int k = 0; k < TH / 4; k++) {
        const int ly = (tid >> 6) + 4 * k;
        const bool in = y0 + ly < h;
        const int y = min(y0 + ly, h - 1);
        const float2 f = sF[ly * TW + lx];
        const size_t o = (size_t)y * w + x;
        if (in) flow[o] = f;
        if (UPDATE) {
            float m[5];
            ffl_um_pixel(R0, R1, plane, w, h, x, y, f.x, f.y, m);
            if (in) {
                float *Mo = Mout + (size_t)b * M_stride + o;
#pragma unroll
                for (int c = 0; c < 5; c++) Mo[c * plane] = m[c];
            }
        }
    }
}

template <int TH>
static void launch_blur_solve_t(const float *Min, float *Mout, size_t M_stride, const float *R, size_t R_stride,
                                size_t plane, PairTab pt, int nB, int lw, int lh, int update, hipStream_t st) {
    dim3 grid(ffl_tile_grid((lw + 63) / 64, (lh + TH - 1) / TH, nB));
    if (update)
        hipLaunchKernelGGL((k_blur_solve<TH, true>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride, plane, pt,
                           lw, lh);
    else
        hipLaunchKernelGGL((k_blur_solve<TH, false>), grid, dim3(256), 0, st, Min, Mout, M_stride, R, R_stride, plane,
                           pt, lw, lh);
}

static int g_blur_tile_h = 16;
void ffl_set_blur_tile_h(int th) { g_blur_tile_h = th; }

void ffl_launch_blur_solve(const float *Min, float *Mout, size_t M_stride, const float *R, size_t R_stride,
                           size_t plane, PairTab pt, int nB, int lw, int lh, int update, hipStream_t st) {
    switch (g_blur_tile_h) {
        case 8: launch_blur_solve_t<8>(Min, Mout, M_stride, R, R_stride, plane, pt, nB, lw, lh, update, st); break;
        default: launch_blur_solve_t<16>(Min, Mout, M_stride, R, R_stride, plane, pt, nB, lw, lh, update, st); break;
    }
}
