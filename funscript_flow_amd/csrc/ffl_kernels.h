// Shared declarations of the gfx950 kernels behind include/ffl.h.
//
// Data layout in HBM (all tightly packed, row-major, sized for level 0 and reused per level):
//   gray   : uint8  [frame slot][h][w]
//   I      : float  [unique frame u][lh][lw]                     level image (blur + resize)
//   R      : float  [unique frame u][5 planes][lh][lw]           polynomial expansion, SoA planes
//   M      : float  [pair b][5 planes][lh][lw]  (two buffers)    G11,G12,G22,h1,h2 before the box blur
//   flow   : float2 [pair b][lh][lw]            interleaved (u,v) exactly as cv2 returns it
// SoA planes (not 20-byte AoS records) so that 64 consecutive lanes read 256 contiguous bytes of
// one channel; the bilinear gather of R1 in UpdateMatrices then hits mostly-shared cache lines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FFL_MAXB 256                 // pairs per batch (= FFL_MAX_BATCH of include/ffl.h)
#define FFL_MAXU (2 * FFL_MAXB)      // unique frames per batch
#define FFL_MAX_LEVELS 4             // pyramid scales (FarnebackOpticalFlowImpl::calc: levels = 3 -> 4 scales)
#define FFL_POLY_N 5
#define FFL_WIN 15
#define FFL_WIN_R 7

struct PolyConsts {
    float g[FFL_POLY_N + 1], xg[FFL_POLY_N + 1], xxg[FFL_POLY_N + 1];
    double gd[FFL_POLY_N + 1], xxgd[FFL_POLY_N + 1];  // (double)g[k], (double)xxg[k]: widened once on the host
    double ig11, ig03, ig33, ig55;
};

struct GaussKernel {
    float k[32];  // full symmetric kernel, ksize taps, centre at k[ksize/2]
    int ksize;
};

// Per-batch tables live in device memory (one copy per compute lane, refreshed by a small stream-ordered H2D
// copy before the batch's first launch): a batch may hold hundreds of pairs (the reference's own operating point
// is 256x256, where only large batches fill the device), far more than fits kernel arguments, and kernels that
// take only pointers and geometry can be replayed from a captured hipGraph.  Every entry is read with a
// wave-uniform index (scalar loads).
struct UTab {  // unique frame u of the batch -> resident frame slot
    int fslot[FFL_MAXU];
};

struct PairTab {  // per pair of the batch
    int u0[FFL_MAXB], u1[FFL_MAXB];          // unique-frame indices of prev / next
    float *flow[FFL_MAX_LEVELS][FFL_MAXB];   // the pair's flow field at every level (level 0: its flow slot);
                                             // level k + 1 is the input of level k's x2 upsample
    struct Pass1Result *res[FFL_MAXB];       // where the pair's pass-1 record goes (mapped pinned memory)
};

struct BatchTab {
    UTab ut;
    PairTab pt;
};

struct Pass1Result {  // written by k_pass1_final, mirrored to pinned host memory
    int x, y;
    float div_val;
    float pad;
    double mag_sum;   // sum of sqrt(u^2+v^2) over the image (mean = mag_sum / (w*h))
};

enum { FFL_FRONT_GENERIC = 0, FFL_FRONT_AREA2 = 1, FFL_FRONT_IDENTITY = 2 };
struct FrontParams {  // k_frontend: decoded 3-channel frame -> gray crop window of its (virtual) resize
    int sw, sh;              // source size
    size_t stride;           // source row pitch in bytes (device copy)
    int cx, cy, ow, oh;      // crop origin inside the resized image, output size
    double scale_x, scale_y; // 1. / ((double)resize / src), formed on the host
    int mode, rgb;
};

// merged launches: one 1-D grid cut into per-job block ranges
#define FFL_MAX_JOBS 4
enum { FFL_PYR_F1 = 0, FFL_PYR_F2, FFL_PYR_H4, FFL_PYR_H9, FFL_PYR_V4, FFL_PYR_V9 };
struct PyrJob {  // one level of the pyramid (caller fills lw, lh, gk, tmp, tmp_stride, I, I_stride)
    int kind, w, h, lw, lh;
    unsigned gx, gy, first, count;  // grid of one frame, first block of the job, tiles of the job (all frames)
    double sx, sy;
    float *tmp, *I;
    size_t tmp_stride, I_stride;
    GaussKernel gk;
};
struct PyrJobs {
    PyrJob j[FFL_MAX_JOBS];
    int n;
};
struct PolyJob {  // one level of PolyExp (caller fills I, I_stride, R, R_stride, plane, w, h)
    const float *I;
    float *R;
    size_t I_stride, R_stride, plane;
    int w, h;
    unsigned gx, gy, first, count;
};
struct PolyJobs {
    PolyJob j[FFL_MAX_JOBS];
    int n;
};

// Tuning knobs (ffl_set_option / ffl_ctx_set_option; results never depend on them).  The process-wide set holds the
// defaults of contexts created afterwards; every context keeps its OWN copy, taken at ffl_create, so two contexts of one
// process (one per GPU, or several on one device) neither share a knob nor invalidate each other's captured graphs.
struct FflOptions {
    int lanes = 2;          // compute lanes (co-scheduled batches) -- fixed once the context exists
    int run_ahead = 0;      // schedule of the frame-only kernels: 0 serial, 1 run-ahead, 2 fork/join
    int fuse_first = 10000; // minimum tiles x pairs of a level for the folded first blur+solve launch (0: never)
    int merge_expand = 1;   // pyramid + PolyExp of all levels in three merged launches
    int use_graph = 1;      // replay a batch's launches from a captured hipGraph
    int copy_threads = 4;   // host threads sharing a staging copy
    int blur_rows = 0;      // tiles a k_blur_solve workgroup walks down (0: automatic)
    int blur_min_wgs = 3500; // automatic strip length: the longest strips that still give this many workgroups
    int tile_order = 0;     // 0 pair-major, 1 tile-major (ffl_tile_coord)
    int pyr_coarse = 1;     // one-pass kernel for the x1/4 and x1/8 pyramid levels where sizes allow
};

// ---- launchers (each enqueues on `st` and returns; no synchronisation) ----------------------
// all pyramid levels in two launches; false (nothing launched) when a level needs the generic kernels
bool ffl_launch_pyr_multi(const uint8_t *gray_base, size_t gray_stride, const UTab *ut, int nU, int w, int h, const PyrJob *levels,
                          int n, const FflOptions &opt, hipStream_t st);
void ffl_launch_polyexp_multi(const PolyJob *levels, int n, int nU, PolyConsts pc, hipStream_t st);
void ffl_launch_frontend(const uint8_t *src, uint8_t *gray, FrontParams p, hipStream_t st);
void ffl_launch_gray(const uint8_t *bgr, uint8_t *gray, int n_pixels, hipStream_t st);
size_t ffl_pyr_tmp_floats(int w, int h, int lw);  // per-frame size of the level's horizontal-pass buffer
void ffl_launch_pyr_level(const uint8_t *gray_base, size_t gray_stride, const UTab *ut, int nU, int w, int h, int lw, int lh,
                          GaussKernel gk, float *tmp, size_t tmp_stride, float *I, size_t I_stride, hipStream_t st);
void ffl_launch_polyexp(const float *I, size_t I_stride, float *R, size_t R_stride, size_t plane, int nU, int lw,
                        int lh, PolyConsts pc, hipStream_t st);
// pw > 0: the level's initial flow = x2 bilinear upsample of pt.prev (pw x ph), used from registers (and written
// to pt.flow only when store_flow != 0: nothing but the debug capture reads it);
// pw == 0: the flow is read from pt.flow, or taken as zero without touching memory when zero_flow != 0
void ffl_launch_update_matrices(const float *R, size_t R_stride, size_t plane, const PairTab *pt, int level, int nB, float *M,
                                size_t M_stride, int lw, int lh, int pw, int ph, int zero_flow, int store_flow,
                                const FflOptions &opt, hipStream_t st);
// update != 0: the next UpdateMatrices is fused in; the solved flow then only reaches memory when store_flow != 0
// (it is dead until the level's last iteration, which always stores it)
void ffl_launch_blur_solve(const float *Min, float *Mout, size_t M_stride, const float *R, size_t R_stride,
                           size_t plane, const PairTab *pt, int level, int nB, int lw, int lh, int update, int store_flow,
                           const FflOptions &opt, hipStream_t st);


void ffl_launch_blur_solve_first(float *Mout, size_t M_stride, const float *R, size_t R_stride, size_t plane,
                                 const PairTab *pt, int level, int nB, int lw, int lh, int pw, int ph, const FflOptions &opt,
                                 hipStream_t st);

int ffl_pass1_blocks(int w, int h);
// pass 1 of the level-0 flows pt->flow[0][b]; records go to pt->res[b]
void ffl_launch_pass1(const PairTab *pt, int nB, int w, int h, int pov_mode, unsigned long long *pkey, double *psum,
                      hipStream_t st);
struct RadialTab {  // device-resident like the batch tables
    const float *flow[FFL_MAXB];
    double cx[FFL_MAXB], cy[FFL_MAXB];
};
// wytab: 2 * h doubles, [y] = (double)(h - y) / h, [h + y] = (double)y / h (the row weights of FF:780-783)
void ffl_launch_radial(const RadialTab *rt, int nB, int w, int h, int pov_mode, const double *wytab, double *psum,
                       double *out, hipStream_t st);

// XCD-aware order of a 1-D run of `count` tiles: the l-th workgroup of the run (l and l+8 share an XCD under the
// observed round-robin placement; the run must start at a multiple of 8) takes tile (l % 8) * chunk + l / 8, so every
// XCD walks one contiguous piece of the run.  ffl_xcd_blocks(count) workgroups cover the run (up to 7 idle ones).
static inline unsigned ffl_xcd_blocks(unsigned count) { return ((count + 7) / 8) * 8; }
__device__ __forceinline__ bool ffl_xcd_tile(unsigned l, unsigned count, unsigned &t) {
    const unsigned chunk = (count + 7) >> 3;
    t = (l & 7u) * chunk + (l >> 3);
    return (l >> 3) < chunk && t < count;
}

// XCD-aware tile order (speed only, never correctness).  Workgroups are dealt round-robin over the 8
// XCDs, so linear ids l and l+8 share an L2.  Each XCD gets one contiguous run of tiles, walked in
// panels of FFL_PANEL_W tile columns, row-major inside a panel: a tile's left/right neighbour runs
// right next to it in time and its upper/lower neighbour FFL_PANEL_W tiles later, all on the same L2,
// so stencil halos and gather neighbourhoods are re-read from L2 instead of over the fabric.
// Grid: ffl_tile_grid(tiles_x, tiles_y, nB) workgroups, 1-D.  Returns false for padding workgroups.
#ifndef FFL_PANEL_W
#define FFL_PANEL_W 4
#endif
static inline unsigned ffl_tile_grid(int tiles_x, int tiles_y, int nB) {
    const int T = tiles_x * tiles_y;
    return (unsigned)(((T + 7) / 8) * 8 * nB);
}
// order 0 (pair-major): all tiles of pair 0, then all tiles of pair 1, ...
// order 1 (tile-major): an XCD walks its run of tiles and runs every tile for ALL nB pairs back to back.  In a
//   stream, frame j's expansion R is R1 of pair j-1 and R0 of pair j: with the same tile of consecutive pairs
//   resident on one XCD at the same time, the second of those reads is served by that XCD's L2 instead of the
//   fabric (the R planes are 40 of the 68 bytes per pixel an UpdateMatrices pass moves).
__device__ __forceinline__ bool ffl_tile_coord(int tiles_x, int tiles_y, int nB, int order, int &b, int &tile_x,
                                               int &tile_y) {
    const int T = tiles_x * tiles_y, chunk = (T + 7) >> 3;
    int t;
    if (order == 0) {
        b = blockIdx.x / (chunk * 8);
        const int l = blockIdx.x - b * (chunk * 8);
        t = (l & 7) * chunk + (l >> 3);
    } else {
        const int pos = blockIdx.x >> 3, tt = pos / nB;
        b = pos - tt * nB;
        t = (blockIdx.x & 7) * chunk + tt;
    }
    if (t >= T) return false;
    const int per_panel = FFL_PANEL_W * tiles_y;
    const int panel = t / per_panel, within = t - panel * per_panel;
    const int pw = min(FFL_PANEL_W, tiles_x - panel * FFL_PANEL_W);  // the last panel may be narrower
    tile_y = within / pw;
    tile_x = panel * FFL_PANEL_W + (within - tile_y * pw);
    return true;
}

// A pointer read from a device-resident table (PairTab / RadialTab) is a generic pointer to the compiler: it
// emits flat_load with per-lane 64-bit address arithmetic and waits on two counters.  Every such pointer is a
// hipMalloc'ed buffer, so its accesses go through these helpers, which name the global address space:
// global_load, saddr form when the base is wave-uniform.  Vectors need 4-byte alignment only.
// Used by the post kernels (k_pass1 -5 %, k_radial -2 %).  NOT by k_blur_solve / k_update_matrices: there the same
// change made the folded launch 3 % slower (1826 -> 1880 us, same-box A/B) -- their coarse-flow gathers stay flat.
#define FFL_GLOBAL __attribute__((address_space(1)))
typedef float ffl_v2f __attribute__((ext_vector_type(2), aligned(4)));
typedef float ffl_v4f __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float2 ffl_gload2(const void *base, size_t byte_off) {
    const ffl_v2f v = *(const FFL_GLOBAL ffl_v2f *)((const char *)base + byte_off);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 ffl_gload4(const void *base, size_t byte_off) {
    const ffl_v4f v = *(const FFL_GLOBAL ffl_v4f *)((const char *)base + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}

// 8- and 16-byte vectors that are only 4-byte aligned: gfx950 global loads/stores of dwordx2/x4 need
// dword alignment only.  What a wave64 global load costs the CU's vector-memory path (cache-resident data, lanes
// contiguous; profiles/tools/micro/vmem_issue.hip): 4 bytes per lane 6.3 cycles, 16 bytes 16.7, 8 bytes 19.5 -- so
// streams use 16-byte accesses (byte taps of the pyramid fetched as words: 54 -> 22 us; PolyExp tiles, flow rows),
// gathers use dwords, and 8-byte loads are avoided (phase V of k_blur_solve with 8-byte loads was slower than with
// dwords; the R1 corner pairs as two dwords instead of one dwordx2: folded launch -3.4 %).
struct __attribute__((packed, aligned(4))) ffl_f2u { float x, y; };
struct __attribute__((packed, aligned(4))) ffl_f4u { float x, y, z, w; };

// Element `idx` of a plane whose base is wave-uniform: base in scalar registers + a 32-bit per-lane byte offset -- the
// global_load / global_store "saddr" form, no per-lane 64-bit address arithmetic (a tenth of UpdateMatrices' vector
// instructions were 64-bit adds).  Offsets fit 32 bits: ffl_create rejects sizes with 20 * w * h >= 2^32.
template <typename T>
__device__ __forceinline__ const T *ffl_at(const float *base, unsigned idx) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + idx * 4u);
}
template <typename T>
__device__ __forceinline__ T *ffl_at(float *base, unsigned idx) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + idx * 4u);
}

// The two horizontally adjacent corners of a bilinear gather / one coarse-flow vector.  A wave64 global load of 8 bytes
// per lane occupies the CU's vector-memory path for ~19 cycles, one of 4 bytes for ~6 and one of 16 bytes for ~17
// (cache-resident data, profiles/tools/micro/vmem_issue.hip): two dword loads are cheaper than one dwordx2.
__device__ __forceinline__ ffl_f2u ffl_ld_corner(const float *plane_base, unsigned idx) {
    ffl_f2u v;
    v.x = *ffl_at<float>(plane_base, idx);
    v.y = *ffl_at<float>(plane_base, idx + 1u);
    return v;
}

// update-matrices body shared by the standalone kernel and the fused blur+solve+update kernel, in three
// steps so that the R1 neighbourhood can be fetched in more than one way:
//   ffl_um_locate   where pixel (x, y) displaced by (dx, dy) lands in R1 and its bilinear weights
//   (gather)        b[c] = a00 * R1c(x1, y1) + a01 * R1c(x1+1, y1) + a10 * R1c(x1, y1+1) + a11 * R1c(x1+1, y1+1)
//   ffl_um_finish   the polynomial-difference terms, border scaling and the 5 products
struct UmLoc {
    int x1, y1;
    float a00, a01, a10, a11;
    bool inside;  // all four corners inside the image
};
__device__ __forceinline__ UmLoc ffl_um_locate(int w, int h, int x, int y, float dx, float dy) {
    UmLoc L;
    float fx = x + dx, fy = y + dy;
    L.x1 = (int)floorf(fx);
    L.y1 = (int)floorf(fy);
    fx -= L.x1;
    fy -= L.y1;
    L.inside = (unsigned)L.x1 < (unsigned)(w - 1) && (unsigned)L.y1 < (unsigned)(h - 1);
    L.a00 = (1.f - fx) * (1.f - fy);
    L.a01 = fx * (1.f - fy);
    L.a10 = (1.f - fx) * fy;
    L.a11 = fx * fy;
    return L;
}

// The inside / outside cases are selects, not a branch: a divergent branch around loaded values makes the compiler wait
// for every outstanding load (vmcnt(0)), the next item's prefetched ones included.
__device__ __forceinline__ void ffl_um_finish(const float (&r0)[5], const float (&b)[5], bool inside, int w, int h, int x,
                                              int y, float dx, float dy, float (&out)[5]) {
    // border[5] = {0.14, 0.14, 0.4472, 0.4472, 0.4472} as selects (no runtime-indexed array)
#define FFL_BORDER(i) ((i) < 2 ? 0.14f : 0.4472f)
    float r2 = inside ? b[0] : 0.f;
    float r3 = inside ? b[1] : 0.f;
    float r4 = inside ? (r0[2] + b[2]) * 0.5f : r0[2];
    float r5 = inside ? (r0[3] + b[3]) * 0.5f : r0[3];
    float r6 = inside ? (r0[4] + b[4]) * 0.25f : r0[4] * 0.5f;
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
        float scale = (x < 5 ? FFL_BORDER(x) : 1.f) * (x >= w - 5 ? FFL_BORDER(w - x - 1) : 1.f) *
                      (y < 5 ? FFL_BORDER(y) : 1.f) * (y >= h - 5 ? FFL_BORDER(h - y - 1) : 1.f);
#undef FFL_BORDER
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    out[0] = r4 * r4 + r6 * r6;
    out[1] = (r4 + r5) * r6;
    out[2] = r5 * r5 + r6 * r6;
    out[3] = r4 * r2 + r6 * r3;
    out[4] = r6 * r2 + r5 * r3;
}
