/*
 * ffl.h -- C ABI of libffl_hip.so: the MI355X (gfx950) implementation of Funscript-Flow's
 * per-frame-pair motion path.  Plain pointers and sizes only; every entry point returns an int
 * status (FFL_OK == 0) unless stated otherwise and records a message readable through
 * ffl_last_error().  There is NO CPU fallback: without a usable HIP device ffl_create() fails.
 *
 * What each entry point replaces in the reference (FF = FunscriptFlow.pyw):
 *
 *   ffl_device_count        availability probe get_available_backends()            FF:32-63
 *   ffl_create/ffl_destroy  (no counterpart: the reference's pair kernel is stateless; the context
 *                           owns the device buffers the CUDA variant allocates per call, FF:984-987)
 *   ffl_upload_frame        the p0/p1 ndarrays handed to precompute_flow_info       FF:843, 1188-1191
 *                           (cuda_GpuMat.upload in the CUDA variant, FF:986-987); also does the
 *                           cv2.cvtColor(..., COLOR_RGB2GRAY) of FF:1082 when given 3 channels
 *   ffl_upload_frames_raw   the decoded frame's way to that operand: cv2.cvtColor(BGR2RGB) FF:182,
 *                           cv2.resize(frame, (256, 256)) FF:185-186 (non-VR) or cv2.resize(f, (512, 512))
 *                           + crop f[256:, :256] FF:1076-1079 (VR), cv2.cvtColor(RGB2GRAY) FF:1079/1082
 *   ffl_flow_pairs          cv2.calcOpticalFlowFarneback(p0,p1,None,0.5,3,15,3,5,1.2,0)  FF:878-879
 *                           + max_divergence(flow)  FF:884 -> FF:748-758
 *                           + cv2.cartToPolar / np.mean                             FF:889-890
 *                           for a whole batch of pairs (Pool.starmap, FF:1190-1191)
 *   ffl_pass1_result        the dict built at FF:898-907 (pos_center, val_pos, mean_mag, cut)
 *   ffl_radial              radial_motion_weighted(flow, center, is_cut, pov_mode)  FF:761-785
 *                           for a batch (ProcessPoolExecutor.submit loop, FF:1232-1236)
 *   ffl_download_flow       the "flow" entry of that dict (tests / callers that want the array)
 *   ffl_submit_pair         precompute_wrapper((p0, p1), params)                    FF:1019-1021
 *
 * Threading: a context is bound to one device and is internally stream-ordered.  Every entry point locks the
 * context, so calls may come from several host threads (an uploader, a submitter and a result collector working
 * on distinct slots).  The per-batch calls do not hold the context lock while they wait for the device or
 * copy frames into staging (ffl_pass1_result(s), ffl_download_flow, ffl_radial, ffl_upload_flow, ffl_sync,
 * ffl_host_free, ffl_upload_frames(_raw) all release it for that time, and wait on events only -- never on a stream
 * another thread may be capturing a graph on); uploads are serialised among themselves, pass-2 calls among themselves.
 * ffl_flow_pairs waits under the lock only when a lane already has 16 batches queued or evicts a captured graph; the
 * test / measurement hooks (ffl_debug_pair, ffl_download_frame, ffl_profile_read) wait under it.
 * ffl_last_error() returns the message of the context's most recent failing call by any thread.
 * Options: ffl_set_option() changes the process-wide DEFAULTS; a context copies them when it is created and is from then
 * on changed only through ffl_ctx_set_option(), so two contexts of one process (one per GPU) share no knob.
 * Sizes: 16x16 <= width x height, 20 * width * height < 2^32 (32-bit plane offsets in the kernels).
 */
#ifndef FFL_H
#define FFL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFL_OK 0
#define FFL_ERR_INVALID 1   /* bad argument (slot out of range, size mismatch, NULL, ...) */
#define FFL_ERR_HIP 2       /* a HIP runtime call failed; see ffl_last_error */
#define FFL_ERR_NO_DEVICE 3 /* no usable gfx950 device */
#define FFL_ERR_STATE 4     /* slot not ready (e.g. result requested before ffl_flow_pairs) */

#define FFL_MAX_BATCH 256   /* pairs per ffl_flow_pairs / ffl_radial call */

typedef struct ffl_ctx ffl_ctx;

/* Number of HIP devices visible to this process (0 when none / no runtime). */
int ffl_device_count(void);

/* Create a context for frames of exactly width x height on `device`.
 *   n_frame_slots  gray frames resident on the device (>= 2)
 *   n_flow_slots   finished flow fields kept resident for pass 2 (a streaming two-pass schedule with two batches
 *                  in flight and the +-6 smoothing window of FF:1203-1214 needs >= 2 * max_batch + 13)
 *   max_batch      pairs processed per ffl_flow_pairs call (1..FFL_MAX_BATCH) */
int ffl_create(int device, int width, int height, int n_frame_slots, int n_flow_slots, int max_batch,
               ffl_ctx **out);
void ffl_destroy(ffl_ctx *ctx);

/* Last error text for ctx (or for the calling thread's failed ffl_create when ctx == NULL). Never NULL.  The text is a
 * per-thread copy taken under the context lock: valid until the calling thread's next ffl_last_error(). */
const char *ffl_last_error(const ffl_ctx *ctx);

/* Sizing a context before creating it (no counterpart in the reference, whose `precomputed` list simply grows in host
 * memory, FF:1191): free / total device memory of `device`, and the device + page-locked host bytes ffl_create would
 * allocate for these arguments under the current "lanes" option.  backend.precompute_all uses them to pick a batch size
 * that fits, or to refuse a chunk that cannot stay resident with a message instead of a failed hipMalloc. */
int ffl_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
int ffl_estimate_bytes(int width, int height, int n_frame_slots, int n_flow_slots, int max_batch, size_t *device_bytes,
                       size_t *pinned_bytes);

/* Copy one frame into frame slot `fslot`.  `channels` is 1 (gray uint8, what the reference feeds
 * Farneback, FF:1082) or 3 (BGR uint8 as cv2.VideoCapture.read returns, FF:178; converted on the
 * device with OpenCV's 8-bit fixed-point luma).  `stride_bytes` is the row pitch of `data`.
 * The pixels are copied into pinned staging before the call returns (the caller may reuse its
 * array); the H2D transfer itself runs on a side stream and overlaps compute already queued. */
int ffl_upload_frame(ffl_ctx *ctx, int fslot, const uint8_t *data, int width, int height, int channels,
                     ptrdiff_t stride_bytes);

/* The same for n frames going to the consecutive slots first_slot .. first_slot+n-1, with one H2D transfer
 * (and one gray-conversion launch) for the whole run: the batched small-image path. */
int ffl_upload_frames(ffl_ctx *ctx, int first_slot, int n, const uint8_t *const *frames, int width, int height,
                      int channels, ptrdiff_t stride_bytes);

/* Input front-end: n decoded 3-channel uint8 frames of src_width x src_height (row pitch
 * stride_bytes; BGR as cv2.VideoCapture.read returns them, or RGB when rgb_order != 0) go to the frame
 * slots first_slot .. first_slot+n-1 as
 *     gray( resize(frame, (resize_width, resize_height)) [crop_y : crop_y+height, crop_x : crop_x+width] )
 * with width x height the context's frame size, cv2.resize's 8-bit INTER_LINEAR rule (11-bit weights; an
 * exact 2x2 down-scale takes INTER_AREA's 2x2 mean; equal sizes skip the resize) and cv2's 8-bit luma
 * (15-bit weights).  The reference's two uses: non-VR (256, 256, 0, 0) on a 256x256 context (FF:185-186,
 * FF:1082); VR (512, 512, 0, 256) on a 256x256 context (FF:1076-1079).  Only the source pixels the crop
 * window samples are read on the device.  Pixels are copied to pinned staging before the call returns. */
int ffl_upload_frames_raw(ffl_ctx *ctx, int first_slot, int n, const uint8_t *const *frames, int src_width,
                          int src_height, ptrdiff_t stride_bytes, int rgb_order, int resize_width, int resize_height,
                          int crop_x, int crop_y);

/* Page-locked host memory owned by the context (freed by ffl_host_free or ffl_destroy).  A decoder that writes its
 * frames into such a buffer saves the library's staging copy: ffl_upload_frame(s) of tightly packed frames that lie
 * back to back inside ONE ffl_host_alloc buffer start the H2D transfer straight out of it.  The caller must then
 * leave those bytes alone until the transfer is over -- after ffl_sync(), or once a result of a batch that uses
 * the frames has been read (ffl_pass1_result).  Frames anywhere else keep the copy-before-return contract.
 * (No counterpart in the reference, whose CUDA variant uploads out of pageable ndarrays, FF:986-987.) */
int ffl_host_alloc(ffl_ctx *ctx, size_t bytes, void **out);
int ffl_host_free(ffl_ctx *ctx, void *ptr);

/* Queue Farneback flow + pass-1 reductions for n pairs: pair i = (frame fslot0[i], frame fslot1[i])
 * -> flow slot flow_slots[i].  Frames shared between pairs of the batch are expanded once.
 * pov_mode != 0 skips the divergence argmax (FF:880-882).  Asynchronous. */
int ffl_flow_pairs(ffl_ctx *ctx, int n, const int *fslot0, const int *fslot1, const int *flow_slots, int pov_mode);

/* Wait for the batch that produced `flow_slot` and return its pass-1 record (FF:898-907):
 * (x, y) = pos_center, div_val = val_pos, mean_mag, cut = mean_mag > cut_threshold. */
int ffl_pass1_result(ffl_ctx *ctx, int flow_slot, float cut_threshold, int32_t *x, int32_t *y, float *div_val,
                     float *mean_mag, int *cut);

/* The same for n slots with one call (arrays of n elements; any output array may be NULL). */
int ffl_pass1_results(ffl_ctx *ctx, int n, const int *flow_slots, float cut_threshold, int32_t *x, int32_t *y,
                      float *div_val, float *mean_mag, int *cut);

/* radial_motion_weighted for n resident flow fields (FF:761-785); out[i] is float64.
 * is_cut[i] != 0 yields 0.0 without touching the device.  Synchronous. */
int ffl_radial(ffl_ctx *ctx, int n, const int *flow_slots, const double *cx, const double *cy, const int *is_cut,
               int pov_mode, double *out);

/* Copy a finished flow field to host memory as (height, width, 2) float32, cv2 layout. */
int ffl_download_flow(ffl_ctx *ctx, int flow_slot, float *dst);

/* Place a caller-provided (height, width, 2) float32 flow field into `flow_slot` and run the pass-1
 * reductions on it (lets the post path be checked against reference golden vectors). */
int ffl_upload_flow(ffl_ctx *ctx, int flow_slot, const float *src, int pov_mode);

/* One-pair convenience: upload prev/next into frame slots 2*slot, 2*slot+1 and queue the pair into
 * flow slot `slot` (mirrors precompute_wrapper, FF:1019-1021). */
int ffl_submit_pair(ffl_ctx *ctx, int slot, const uint8_t *prev, const uint8_t *next, int width, int height,
                    int channels, ptrdiff_t stride_bytes, int pov_mode);

/* Block until everything queued on the context has finished. */
int ffl_sync(ffl_ctx *ctx);

/* ---- parity-test hooks (used by tests/ only) ------------------------------------------------ */

/* Number of pyramid scales minus one for this context's size (3 for every BASELINE config). */
int ffl_num_levels(const ffl_ctx *ctx);
/* Level-k geometry: out_wh[0]=width, out_wh[1]=height. */
int ffl_level_size(const ffl_ctx *ctx, int level, int *out_wh);

/* Copy the resident gray frame of `fslot` to host memory (height * width bytes). */
int ffl_download_frame(ffl_ctx *ctx, int fslot, uint8_t *dst);

/* Run ONE pair (frame slots f0, f1) and capture the level-`level` intermediates as they stand
 * before blur iteration `iter` (0: right after the initial UpdateMatrices; 3: end of level).
 * Any of the output pointers may be NULL.  Planar layouts: R*, M = 5 planes of lh*lw floats;
 * I* = lh*lw; flow = lh*lw*2 interleaved.  The final full-resolution flow goes to flow slot 0. */
int ffl_debug_pair(ffl_ctx *ctx, int f0, int f1, int level, int iter, float *I0, float *I1, float *R0, float *R1,
                   float *M, float *flow);

/* ---- measurement hooks (bench.py) -------------------------------------------------------------- */

/* Tuning knobs (results never depend on them).  ffl_set_option sets the process-wide default that contexts created
 * AFTERWARDS start from; ffl_ctx_set_option changes one live context (every knob but "lanes", which sizes the context's
 * buffers: FFL_ERR_STATE) and makes that context -- and no other -- re-capture its graphs; ffl_ctx_get_option reads a
 * context's value (ctx == NULL: the process-wide default).
 *   "blur_tile_h" = 16       rows of the 64-wide k_blur_solve LDS tile.  Fixed since the box-sum order
 *                            is anchored to blocks of 16 rows / columns (other values are refused); the
 *                            8 / 16 / 32 sweep of BASELINE configs[2] is recorded in profiles/README.md
 *   "blur_rows"   = 0..64    tiles a k_blur_solve workgroup walks down (0 = automatic, the default)
 *   "blur_min_wgs" = N >= 1  automatic strip length: a column of tiles is cut into the fewest equal strips that still
 *                            give the launch N workgroups (default 3500)
 *   "fuse_first"  = N >= 0   a level's flow init + first UpdateMatrices run inside its first k_blur_solve launch
 *                            when the level has at least N 64x16 tiles over the batch (default 10000; 0 = never,
 *                            1 = always)
 *   "merge_expand" = 0|1     1 (default): pyramid + PolyExp of all levels in three merged launches per batch;
 *                            0: one set of launches per level
 *   "lanes"       = 1..4     compute lanes (co-scheduled batches, each on its own stream and work
 *                            buffers) of contexts created afterwards; default 2
 *   "run_ahead"   = 0|1|2    schedule of the frame-only kernels: 0 serial (default), 1 run-ahead on a
 *                            side stream, 2 fork/join over per-level side streams
 *   "copy_threads" = 1..16   host threads that share a staging copy of 1 MiB or more (the caller + helpers owned by
 *                            the context); default 4.  One thread moves ~22 GB/s, a 1080p BGR stream at 5 k pairs/s
 *                            needs 31
 *   "graph"       = 0|1      1 (default): a batch's launches are captured once per (lane, batch shape, option set)
 *                            into a hipGraph and replayed; 0: launched one by one (timing events and the debug
 *                            capture always launch one by one)
 *   "pyr_coarse"  = 0|1      1 (default): the x1/4 and x1/8 pyramid levels of frames whose sides are multiples of 8
 *                            come from one LDS-staged pass (k_pyr_coarse); 0: horizontal + vertical kernel pairs
 *   "tile_order"  = 0|1      k_blur_solve / k_update_matrices workgroup order: 0 pair-major (default), 1 tile-major
 *                            (every tile for all pairs of the batch back to back; less fabric traffic, not faster) */
int ffl_set_option(const char *name, int value);
int ffl_ctx_set_option(ffl_ctx *ctx, const char *name, int value);
int ffl_ctx_get_option(ffl_ctx *ctx, const char *name, int *value);

/* hipGraph bookkeeping of a context: batch shapes captured, batches replayed from a graph, captures that FAILED.  A
 * failed capture is never silent: the batch is launched kernel by kernel (results unaffected), the context stops
 * capturing until one of its options changes, one line goes to stderr, and the failure is counted here -- bench.py
 * reports the three numbers (`config.graphs`) and the GPU tests assert capture_failures == 0. */
int ffl_graph_stats(ffl_ctx *ctx, int *captured, int *replayed, int *capture_failures);

/* HIP-event timing of kernel classes: every launch of a class whose bit (1u << FFL_K_*) is set in
 * class_mask is bracketed by events on the stream it is launched on.  0 switches timing off. */
int ffl_profile_enable(ffl_ctx *ctx, unsigned class_mask);
#define FFL_PROFILE_ALL 0xFFu
#define FFL_K_GRAY 0
#define FFL_K_PYRAMID 1
#define FFL_K_POLYEXP 2
#define FFL_K_FRONTEND 3 /* k_frontend of ffl_upload_frames_raw (the x2 flow upsample, once class 3, runs
                            inside k_update_matrices) */
#define FFL_K_UPDATE_MATRICES 4
#define FFL_K_BLUR_SOLVE 5
#define FFL_K_PASS1 6
#define FFL_K_RADIAL 7
#define FFL_K_COUNT 8
/* Read and reset the accumulated (launch count, total milliseconds) of one kernel class.
 * Synchronises the context. */
int ffl_profile_read(ffl_ctx *ctx, int kernel_class, int *launches, double *total_ms);
const char *ffl_kernel_name(int kernel_class);

#ifdef __cplusplus
}
#endif
#endif /* FFL_H */
