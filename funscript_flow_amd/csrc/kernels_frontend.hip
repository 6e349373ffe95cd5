// Input front-end: decoded 3-channel u8 frame -> the gray operand of the pair
// kernel, in one pass.  Replaces, for the HIP backend, the reference's per-frame host work
//   cv2.cvtColor(BGR2RGB) FF:182, cv2.resize(frame, (256, 256)) FF:185-186 / cv2.resize(f, (512, 512)) +
//   crop f[256:, :256] FF:1076-1079, cv2.cvtColor(RGB2GRAY) FF:1079/1082.
// Every output pixel of the crop window is computed directly from its (up to) 4 source pixels with
// OpenCV's 8-bit fixed-point rules (11-bit lerp weights, 15-bit luma weights) -- integer work, so the
// result is bit-identical to the two-pass CPU restatement in oracle/frontend_oracle.c.
// Roofline: HBM / PCIe -- the kernel touches at most 12 source bytes per output pixel; the frame's
// H2D transfer (3 * src_w * src_h bytes) is what bounds the path.
#include "ffl_kernels.h"

__device__ __forceinline__ int ffl_sat_short_round(float v) {
    int r = (int)rintf(v);  // round half to even (cvRound)
    return min(max(r, -32768), 32767);
}

__global__ __launch_bounds__(256) void k_frontend(const uint8_t *__restrict__ src, uint8_t *__restrict__ gray,
                                                  FrontParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.ow || y >= p.oh) return;
    const int dx = x + p.cx, dy = y + p.cy;  // position in the (virtual) resized image
    int v[3];
    if (p.mode == FFL_FRONT_IDENTITY) {
        const uint8_t *s = src + (size_t)dy * p.stride + 3 * dx;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
    } else if (p.mode == FFL_FRONT_AREA2) {  // exact 2x2 down-scale: INTER_LINEAR is routed to INTER_AREA
        const uint8_t *s0 = src + (size_t)(2 * dy) * p.stride + 6 * dx, *s1 = s0 + p.stride;
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = (s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2;
    } else {
        float fx = (float)((dx + 0.5) * p.scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= p.sw - 1) { sx = p.sw - 1; fx = 0.f; }
        const int sx1 = min(sx + 1, p.sw - 1);
        const int a0 = ffl_sat_short_round((1.f - fx) * 2048.f), a1 = ffl_sat_short_round(fx * 2048.f);
        float fy = (float)((dy + 0.5) * p.scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        const int b0 = ffl_sat_short_round((1.f - fy) * 2048.f), b1 = ffl_sat_short_round(fy * 2048.f);
        const int y0 = min(max(sy, 0), p.sh - 1), y1 = min(max(sy + 1, 0), p.sh - 1);
        const uint8_t *S0 = src + (size_t)y0 * p.stride, *S1 = src + (size_t)y1 * p.stride;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = S0[3 * sx + c] * a0 + S0[3 * sx1 + c] * a1;
            const int h1 = S1[3 * sx + c] * a0 + S1[3 * sx1 + c] * a1;
            v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        }
    }
    const int r = p.rgb ? v[0] : v[2], b = p.rgb ? v[2] : v[0];
    gray[(size_t)y * p.ow + x] = (uint8_t)((r * 9798 + v[1] * 19235 + b * 3735 + 16384) >> 15);
}

void ffl_launch_frontend(const uint8_t *src, uint8_t *gray, FrontParams p, hipStream_t st) {
    dim3 grid((p.ow + 63) / 64, (p.oh + 3) / 4);
    hipLaunchKernelGGL(k_frontend, grid, dim3(256), 0, st, src, gray, p);
}
