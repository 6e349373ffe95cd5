// C ABI (include/ffl.h) of the gfx950 pair-motion path: context, slots, streams, batch schedule.
//
// Streams: `copy` carries the pinned H2D frame transfers (+ the BGR->gray kernel); every compute lane
// has its own stream (+ side streams for the optional run-ahead schedules) and work buffers; `post`
// carries pass 2.  A batch waits on its frames' upload events; an upload into a frame slot waits on
// the batch events of the lanes that last read it; each batch records ONE event that stands for "slots
// ready / frames released / lane buffers free".  Result records are stored by the reduction kernels
// straight into mapped pinned memory, so no tiny D2H copies are queued.  No host synchronisation
// happens inside ffl_upload_frame / ffl_flow_pairs, so uploads of the next frames overlap the kernels
// of the previous batch (north_star: "staged to HBM via pinned hipMemcpyAsync on a side stream").
#include "../../include/ffl.h"
#include "ffl_kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define FFL_EV_RING 16
#define FFL_RAW_RING 4

static thread_local std::string g_create_error = "";
// The process-wide option set (ffl_set_option): the defaults of contexts created AFTERWARDS.  Every context copies it at
// ffl_create (ffl_ctx::opt) and is from then on only changed through ffl_ctx_set_option, which bumps that context's own
// graph epoch -- two contexts of one process (one per GPU) share no knob and do not invalidate each other's graphs.
//   run_ahead   schedule of the frame-only kernels (see enqueue_batch).  Measured at 1080p, B = 8 (pairs/s with 1 / 2
//               lanes): 0 serial 3840 / 4063, 2 fork-join 3641 / 4101, 1 run-ahead 3943 / 4159.
//   fuse_first  a level's initial UpdateMatrices runs inside its first blur+solve launch when the level has at least this
//               many 64x16 tiles over the batch (0: never).  The folded launch saves an M write + read but runs its extra
//               phase at 3 workgroups per CU: worth it only where the launch is long enough to be bandwidth-bound (1080p,
//               B = 32: level 0 2093 vs 1018 + 1217 us; level 2 157 vs 65 + 68 us; level 3 138 vs 34 + 45 us)
//   use_graph   a batch's ~20 launches are captured once per (lane, batch shape, option epoch) into a hipGraph and
//               replayed: the kernels take only pointers and geometry (per-batch indices live in the lane's device table),
//               so nothing in the graph changes from batch to batch.  Host time per batch drops from one launch call per
//               kernel to one graph launch -- what matters at the reference's 256x256 operating point, where a whole
//               batch is a fraction of a ms.
static FflOptions g_opts;
static std::mutex g_opt_mu;

// Events come from small rings, one per stream (a lane's "batch finished" events, the upload events of stream `copy`, the
// events of stream `post`), and what the slot tables hold are REFERENCES to ring entries: (ring, ticket).  An entry is
// recorded again once the ring has gone round.  A reference that outlived its entry must not be waited on any more: the
// event it names now stands for a much LATER operation of that stream -- possibly one that is still running -- and waiting
// for it serialises things that have nothing to do with each other.  (Rounds 1-3 kept bare hipEvent_t handles.  Harmless
// for correctness, since "later work of the same stream" is a conservative wait, but not for speed: with 131 or 132 frame
// slots at B = 32 a slot's last-use handle of the OTHER lane went 33 batches without being refreshed, and from the moment
// the lanes' 16-entry rings wrapped every upload waited for the batch in flight: 5.1 -> 8.9 ms per batch from batch 33 on,
// profiles/r04_pcie_chunk_length.txt.)  So: take() waits on the host for the entry it is about to hand out again -- that
// operation is `size` operations old and practically always long finished -- which makes "stale" imply "completed", and
// EvRef::get() answers nullptr for a stale reference: nothing to wait for.
struct EvRing {
    std::vector<hipEvent_t> ev;
    unsigned long long next = 0;  // tickets handed out so far
    hipError_t create(int n) {
        ev.assign(n, nullptr);
        for (auto &e : ev) {
            hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    }
    void destroy() {
        for (auto e : ev)
            if (e) hipEventDestroy(e);
        ev.clear();
    }
    // the entry the next take() hands out: its previous operation (`size` operations ago) must be over
    hipError_t settle_next() { return next >= ev.size() ? hipEventSynchronize(ev[next % ev.size()]) : hipSuccess; }
    hipEvent_t take(unsigned long long *ticket) {
        *ticket = next;
        return ev[next++ % ev.size()];
    }
};
struct EvRef {
    const EvRing *ring = nullptr;
    unsigned long long ticket = 0;
    // the event to wait on, or nullptr when there is nothing (never set, or the entry has been handed out again: completed)
    hipEvent_t get() const { return (ring && ring->next - ticket <= ring->ev.size()) ? ring->ev[ticket % ring->ev.size()] : nullptr; }
};
static inline EvRef ev_latest(const EvRing &r) { return r.next ? EvRef{&r, r.next - 1} : EvRef{}; }

struct ProfRec {
    int cls;
    hipEvent_t a, b;
    bool shared_start;  // `a` is the previous record's `b`
};

struct LevelGeom {
    int lw, lh, ksize;
    double sigma;
    GaussKernel gk;
};

// Staging copies (caller's pageable ndarray -> pinned memory) bound the PCIe-inclusive rate from 3-channel frames:
// one host thread moves ~22 GB/s, a 1080p BGR stream at 5 k pairs/s needs 31.  A few persistent helper threads share
// each large copy by rows (created on first use, joined in ffl_destroy).
struct CopyPool {
    // one share of a staging copy: global rows g0 .. g1-1 of a run of `nf` equally shaped frames (row g belongs to frame
    // g / rows); frame f is read from srcs[f] (row pitch src_pitch) and lands at dst + f * dst_frame (rows packed)
    struct Job {
        uint8_t *dst;
        const uint8_t *const *srcs;
        size_t row_bytes, dst_frame;
        ptrdiff_t src_pitch;
        int rows, g0, g1;
    };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Job> jobs;   // one slot per worker
    std::vector<char> busy;
    bool stop = false;

    static void run(const Job &j) {
        int g = j.g0;
        while (g < j.g1) {
            const int f = g / j.rows, y0 = g - f * j.rows;
            const int y1 = (j.g1 - f * j.rows) < j.rows ? (j.g1 - f * j.rows) : j.rows;  // rows y0 .. y1-1 of frame f
            uint8_t *d = j.dst + (size_t)f * j.dst_frame + (size_t)y0 * j.row_bytes;
            const uint8_t *s = j.srcs[f] + (ptrdiff_t)y0 * j.src_pitch;
            if (j.src_pitch == (ptrdiff_t)j.row_bytes)
                memcpy(d, s, j.row_bytes * (size_t)(y1 - y0));
            else
                for (int y = y0; y < y1; y++, d += j.row_bytes, s += j.src_pitch) memcpy(d, s, j.row_bytes);
            g += y1 - y0;
        }
    }
    void start(int n) {
        jobs.resize(n);
        busy.assign(n, 0);
        for (int i = 0; i < n; i++)
            workers.emplace_back([this, i] {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    cv_work.wait(lk, [&] { return stop || busy[i]; });
                    if (stop) return;
                    Job j = jobs[i];
                    lk.unlock();
                    run(j);
                    lk.lock();
                    busy[i] = 0;
                    cv_done.notify_all();
                }
            });
    }
    // `nf` frames of rows x row_bytes (source row pitch src_pitch) into consecutive packed frames at dst, the run's rows
    // split evenly over the helpers + the caller.  A run of many SMALL frames (257 frames of 64 KB per batch at the
    // reference's 256x256 operating point) is shared like one large frame: copied frame by frame on the calling thread it
    // cost 0.7 ms per batch -- half of what the device needs for the batch -- and made the host the bottleneck.
    void copy(uint8_t *dst, const uint8_t *const *srcs, int nf, ptrdiff_t src_pitch, size_t row_bytes, int rows, int threads) {
        const int total = nf * rows;
        const int parts = (row_bytes * (size_t)total < (size_t)(1 << 20) || threads < 2) ? 1 : (threads < total ? threads : total);
        if (parts > 1 && (int)workers.size() < parts - 1) {
            // (re)size once; helpers are idle here because copy() is only called under the context's upload lock (up_mu)
            shutdown();
            stop = false;
            start(parts - 1);
        }
        const int per = (total + parts - 1) / parts;
        const size_t dst_frame = row_bytes * (size_t)rows;
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu);
            for (int p = 1; p < parts; p++) {
                const int g0 = p * per, g1 = g0 + per < total ? g0 + per : total;
                if (g0 >= g1) continue;
                jobs[p - 1] = {dst, srcs, row_bytes, dst_frame, src_pitch, rows, g0, g1};
                busy[p - 1] = 1;
            }
            cv_work.notify_all();
        }
        run({dst, srcs, row_bytes, dst_frame, src_pitch, rows, 0, per < total ? per : total});
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] {
                for (char b : busy)
                    if (b) return false;
                return true;
            });
        }
    }
    void shutdown() {
        {
            std::unique_lock<std::mutex> lk(mu);
            stop = true;
            cv_work.notify_all();
        }
        for (auto &t : workers) t.join();
        workers.clear();
    }
};

struct ffl_ctx {
    int device = 0, w = 0, h = 0, levels = 0;
    FflOptions opt;      // this context's own option set (copied from the process-wide defaults at ffl_create)
    int opt_epoch = 0;   // bumped by every applied ffl_ctx_set_option: a graph is only replayed under the options it was captured with
    int graph_captured = 0, graph_replayed = 0, graph_failed = 0;  // ffl_graph_stats
    CopyPool pool;
    int n_fslots = 0, n_slots = 0, max_batch = 0;
    size_t N = 0;
    LevelGeom geom[8];
    PolyConsts pc;
    hipStream_t s_copy = nullptr, s_post = nullptr;  // uploads (+gray) / pass 2 and flow uploads
    // Compute lanes: each ffl_flow_pairs batch runs on the next lane (stream + its own work buffers),
    // so consecutive batches execute concurrently and the device overlaps one batch's f64-bound box
    // filter with another's bandwidth-bound UpdateMatrices / PolyExp.
    struct Lane {
        hipStream_t st = nullptr;
        // Frame-only work (pyramid + PolyExp of every level) runs ahead on `st_aux` and overlaps the
        // flow chain of the coarser levels, whose small grids leave most of the device idle; the
        // chain on `st` waits for ev_R[k] before touching level k.  R holds all levels at once.
        hipStream_t st_aux[4] = {nullptr};
        hipEvent_t ev_R[8] = {nullptr}, ev_fork = nullptr;
        size_t i_off[8] = {0};  // float offset of level k inside d_I
        size_t t_off[8] = {0};  // float offset of level k inside d_T (pyramid horizontal-pass buffer)
        float *d_T = nullptr;
        EvRing ring;  // one "batch finished" event per batch (FFL_EV_RING entries)
        size_t r_off[8] = {0};  // float offset of level k inside d_R
        float *d_I = nullptr, *d_R = nullptr, *d_M[2] = {nullptr, nullptr}, *d_flowA = nullptr, *d_flowB = nullptr;
        unsigned long long *d_pkey = nullptr;
        double *d_psum = nullptr;
        // per-batch index tables: device copy + a ring of pinned host copies (entry e belongs to ring entry e)
        BatchTab *d_tab = nullptr, *h_tab = nullptr;
        struct GraphEntry {
            int n, nU, pov, epoch;
            hipGraph_t graph;
            hipGraphExec_t exec;
        };
        std::vector<GraphEntry> graphs;
    };
    std::vector<Lane> lanes;
    unsigned next_lane = 0;
    // frames
    uint8_t *d_gray = nullptr;        // [n_fslots][N]
    uint8_t *d_bgr = nullptr;         // [n_fslots][3N] staging for 3-channel uploads
    uint8_t *h_stage_gray = nullptr;  // pinned [n_fslots][N]   (separate, so that runs of slots are contiguous)
    uint8_t *h_stage_bgr = nullptr;   // pinned [n_fslots][3N]
    std::vector<EvRef> ev_uploaded;  // per frame slot: the upload call that filled it (one event per call, up_ring)
    EvRing up_ring;                  // 2 * FFL_EV_RING entries
    std::vector<EvRef> ev_last_use;  // [frame slot * n_lanes + lane]: the last batch of that lane that read the slot
    std::vector<char> frame_valid;
    std::vector<int> u_of_fslot;      // scratch of run_batch: frame slot -> index among the batch's unique frames (-1 outside)
    std::vector<char> slot_mark;      // scratch of check_pairs: flow slot already named in this batch
    // ffl_upload_frames_raw: decoded source frames pass through a small ring of pinned + device buffers
    // (grown on demand to the largest source seen); `ev` = the frame's k_frontend has consumed the buffer
    struct RawBuf {
        uint8_t *h = nullptr, *d = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool busy = false;
    };
    RawBuf raw[FFL_RAW_RING];
    unsigned raw_next = 0;
    EvRing post_ring;  // events of ffl_upload_flow and ffl_radial (stream `post`), FFL_EV_RING entries
    // flow slots
    float *d_flow = nullptr;          // [n_slots][2N]
    // Result records live in pinned, device-mapped host memory: the reduction kernels store their
    // 24-byte record straight into it (visible after the slot's event), so no D2H copies are queued.
    Pass1Result *h_res = nullptr;     // pinned [n_slots]
    Pass1Result *d_res = nullptr;     // device alias of h_res
    std::vector<EvRef> ev_slot_done;  // per flow slot: the batch (a lane's ring) or pass-2 / flow-upload call (post_ring) that used it last
    std::vector<char> slot_state;     // 0 empty, 1 queued/ready
    std::vector<char> slot_pov;
    RadialTab *d_rtab = nullptr, *h_rtab = nullptr;    // pass-2 table (s_post; ffl_radial waits for the stream, so one copy)
    BatchTab *d_ptab = nullptr, *h_ptab = nullptr;     // ffl_upload_flow's one-pair table (s_post)
    double *d_rpsum = nullptr;                         // pass-2 partial sums (s_post)
    double *d_wytab = nullptr;                         // pass-2 row weights (h - y) / h and y / h
    double *h_radial = nullptr, *d_radial = nullptr;   // pinned pass-2 results and their device alias
    unsigned long long *d_ppkey = nullptr;                                 // ffl_upload_flow scratch (s_post)
    int p1_blocks = 0;
    // profiling
    unsigned prof_mask = 0;   // bit k set: bracket every launch of kernel class k with HIP events
    // caller-visible page-locked buffers (ffl_host_alloc): uploads out of them skip the staging copy
    std::vector<std::pair<uint8_t *, size_t>> host_bufs;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;   // recycled timing events
    hipEvent_t prof_last_end = nullptr;  // end event of the latest timed launch, while nothing followed it
    int prof_last_cls = -1;
    hipStream_t prof_last_stream = nullptr;
    int prof_launches[FFL_K_COUNT] = {0};
    double prof_ms[FFL_K_COUNT] = {0};
    std::string err;
    // Every entry point takes this lock, so calls from several host threads are safe.  The calls a pipeline makes per
    // batch do not hold it while they wait for the device or copy frames: ffl_pass1_result, ffl_download_flow, ffl_radial,
    // ffl_upload_flow, ffl_sync and ffl_host_free drop it around their waits, ffl_upload_frames(_raw) around the staging
    // memcpy -- so a thread collecting results does not hold up another one that is uploading frames or queueing the next
    // batch (submit / pass1 / radial on distinct slots may come from different host threads).  Exceptions, all
    // rare or measurement-only: ffl_flow_pairs waits under the lock when a lane has FFL_EV_RING batches queued (its table
    // ring is full) or evicts a captured graph (the 17th batch shape of a lane); ffl_debug_pair and ffl_download_frame
    // are test hooks and wait under it; ffl_profile_read collects its timing events under it.
    // A wait that runs WITHOUT the lock only ever waits on EVENTS, never on a lane's stream: another thread may be inside
    // hipStreamBeginCapture / EndCapture on that stream (a new batch shape), and synchronising a capturing stream is an
    // error that also invalidates the capture; hipEventSynchronize on an event recorded outside the capture is legal.
    // Two small locks order the users of shared single-copy resources among themselves; both are taken BEFORE `mu`, and
    // up_mu before post_mu where a call needs both (ffl_sync):
    //   up_mu    uploaders: the per-slot staging areas, the copy pool and the raw-frame ring
    //   post_mu  users of stream `post` and its single pinned tables / result buffer (ffl_radial, ffl_upload_flow)
    mutable std::recursive_mutex mu;
    std::mutex up_mu, post_mu;
    int graph_bad_epoch = -1;  // option epoch in which a graph capture failed: batches launch eagerly until it changes
    bool graph_fail_reported = false;
};
typedef std::unique_lock<std::recursive_mutex> CtxLock;

static int set_err(ffl_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return set_err(c, FFL_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                                     \
    } while (0)

// ---- host-side constants (same published procedure as OpenCV's helpers) -------------------------
static inline int cv_round(double v) { return (int)lrint(v); }

static void gaussian_kernel(int n, double sigma, float *out) {  // getGaussianKernel(n, sigma, CV_32F)
    static const float tab1[] = {1.f};
    static const float tab3[] = {0.25f, 0.5f, 0.25f};
    static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    const float *fixed = nullptr;
    if (sigma <= 0 && (n & 1) && n <= 7) fixed = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
    double sg = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sg * sg);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2x * x * x);
        out[i] = (float)t;
        sum += out[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) out[i] = (float)(out[i] * sum);
}

static void polyexp_prepare(PolyConsts *pc) {  // FarnebackPrepareGaussian(n = 5, sigma = 1.2)
    const int n = FFL_POLY_N;
    const double sigma = 1.2;
    float gg[2 * FFL_POLY_N + 1];
    double s = 0;
    for (int x = -n; x <= n; x++) {
        gg[x + n] = (float)exp(-x * x / (2 * sigma * sigma));
        s += gg[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) gg[x + n] = (float)(gg[x + n] * s);
    for (int x = 0; x <= n; x++) {
        pc->g[x] = gg[x + n];
        pc->xg[x] = (float)(x * gg[x + n]);
        pc->xxg[x] = (float)(x * x * gg[x + n]);
        pc->gd[x] = (double)pc->g[x];
        pc->xxgd[x] = (double)pc->xxg[x];
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            float p = gg[y + n] * gg[x + n];
            G[0][0] += p;
            G[1][1] += p * x * x;
            G[3][3] += p * x * x * x * x;
            G[5][5] += p * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double A[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 12; j++) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 12; j++) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
        double d = 1.0 / A[c][c];
        for (int j = 0; j < 12; j++) A[c][j] *= d;
        for (int r = 0; r < 6; r++)
            if (r != c) {
                double f = A[r][c];
                if (f != 0)
                    for (int j = 0; j < 12; j++) A[r][j] -= f * A[c][j];
            }
    }
    pc->ig11 = A[1][7];
    pc->ig03 = A[0][9];
    pc->ig33 = A[3][9];
    pc->ig55 = A[5][11];
}

static void level_geometry(ffl_ctx *c) {  // FarnebackOpticalFlowImpl::calc level logic
    int k;
    double scale = 1.0;
    for (k = 0; k < 3; k++) {
        scale *= 0.5;
        if (c->w * scale < 32 || c->h * scale < 32) break;
    }
    c->levels = k;
    for (k = 0; k <= c->levels; k++) {
        double sc = 1.0;
        for (int i = 0; i < k; i++) sc *= 0.5;
        LevelGeom &g = c->geom[k];
        g.sigma = (1.0 / sc - 1.0) * 0.5;
        int sm = cv_round(g.sigma * 5) | 1;
        g.ksize = sm < 3 ? 3 : sm;
        g.lw = cv_round(c->w * sc);
        g.lh = cv_round(c->h * sc);
        memset(&g.gk, 0, sizeof(g.gk));
        g.gk.ksize = g.ksize;
        gaussian_kernel(g.ksize, g.sigma, g.gk.k);
    }
}

// ---- profiling helpers ---------------------------------------------------------------------------
// Timing events come from a per-context pool (creating two events per launch cost more host time than the
// launch itself).  Where the caller knows that timed launches of a class are queued back to back (the three
// k_blur_solve iterations of a level) a launch starts at its predecessor's end event instead of recording one.
static hipEvent_t prof_event(ffl_ctx *c) {
    if (!c->prof_pool.empty()) {
        hipEvent_t e = c->prof_pool.back();
        c->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

struct ProfScope {
    ffl_ctx *c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    bool on, shared = false;
    // chain: the caller guarantees that nothing was queued on `st` since the previous timed launch of this class
    ProfScope(ffl_ctx *c_, int cls_, hipStream_t st, bool chain = false)
        : c(c_), cls(cls_), on((c_->prof_mask >> cls_) & 1u) {
        if (on) {
            stream = st;
            if (chain && c->prof_last_end && c->prof_last_cls == cls && c->prof_last_stream == st) {
                a = c->prof_last_end;  // nothing was queued on `st` since that launch ended
                shared = true;
            } else {
                a = prof_event(c);
                hipEventRecord(a, st);
            }
        } else {
            c->prof_last_end = nullptr;  // an untimed launch breaks the chain
        }
    }
    ~ProfScope() {
        if (on) {
            b = prof_event(c);
            hipEventRecord(b, stream);
            c->prof_recs.push_back({cls, a, b, shared});
            c->prof_last_end = b;
            c->prof_last_cls = cls;
            c->prof_last_stream = stream;
        }
    }
    hipStream_t stream = nullptr;
};

static void prof_collect(ffl_ctx *c) {
    for (auto &r : c->prof_recs) {
        float ms = 0;
        hipEventSynchronize(r.b);
        hipEventElapsedTime(&ms, r.a, r.b);
        c->prof_launches[r.cls]++;
        c->prof_ms[r.cls] += ms;
    }
    for (auto &r : c->prof_recs) {
        if (!r.shared_start) c->prof_pool.push_back(r.a);
        c->prof_pool.push_back(r.b);
    }
    c->prof_recs.clear();
    c->prof_last_end = nullptr;
}

// hipStreamWaitEvent for every DISTINCT event of a list: the 257 frames of a 256-pair batch share one or two upload events and
// its 256 recycled flow slots one or two batch events, and a wait call costs ~0.5 us each (0.25 ms per 256-pair batch).
struct WaitOnce {
    hipStream_t st;
    hipEvent_t seen[8];
    int n = 0;
    explicit WaitOnce(hipStream_t s) : st(s) {}
    hipError_t operator()(hipEvent_t e) {
        if (!e) return hipSuccess;
        for (int k = 0; k < n; k++)
            if (seen[k] == e) return hipSuccess;
        if (n < 8) seen[n++] = e;
        return hipStreamWaitEvent(st, e, 0);
    }
};

// ---- API -----------------------------------------------------------------------------------------
extern "C" {

int ffl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ffl_last_error(const ffl_ctx *ctx) {
    if (!ctx) return g_create_error.c_str();
    // copied under the lock into a per-thread buffer: another thread's failing call may replace ctx->err while the caller
    // still reads the text (valid until this thread's next ffl_last_error call)
    static thread_local std::string copy;
    CtxLock lk(ctx->mu);
    copy = ctx->err;
    return copy.c_str();
}

const char *ffl_kernel_name(int k) {
    static const char *names[FFL_K_COUNT] = {"k_gray",  "k_pyr_level", "k_polyexp", "k_frontend",
                                             "k_update_matrices", "k_blur_solve", "k_pass1", "k_radial"};
    return (k >= 0 && k < FFL_K_COUNT) ? names[k] : "?";
}

void ffl_destroy(ffl_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    for (auto &L : c->lanes)
        if (L.st) hipStreamSynchronize(L.st);
    if (c->s_post) hipStreamSynchronize(c->s_post);
    if (c->s_copy) hipStreamSynchronize(c->s_copy);
    prof_collect(c);
    c->pool.shutdown();
    for (auto e : c->prof_pool) hipEventDestroy(e);
    for (auto &hb : c->host_bufs) hipHostFree(hb.first);
    c->up_ring.destroy();
    c->post_ring.destroy();
    hipFree(c->d_gray); hipFree(c->d_bgr); hipHostFree(c->h_stage_gray); hipHostFree(c->h_stage_bgr);
    for (auto &rb : c->raw) {
        hipFree(rb.d);
        hipHostFree(rb.h);
        if (rb.ev) hipEventDestroy(rb.ev);
    }
    for (auto &L : c->lanes) {
        hipFree(L.d_I); hipFree(L.d_T); hipFree(L.d_R); hipFree(L.d_M[0]); hipFree(L.d_M[1]);
        hipFree(L.d_flowA); hipFree(L.d_flowB); hipFree(L.d_pkey); hipFree(L.d_psum);
        hipFree(L.d_tab); hipHostFree(L.h_tab);
        for (auto &g : L.graphs) {
            hipGraphExecDestroy(g.exec);
            hipGraphDestroy(g.graph);
        }
        for (auto e : L.ev_R)
            if (e) hipEventDestroy(e);
        L.ring.destroy();
        if (L.ev_fork) hipEventDestroy(L.ev_fork);
        for (auto s : L.st_aux)
            if (s) hipStreamDestroy(s);
        if (L.st) hipStreamDestroy(L.st);
    }
    hipFree(c->d_flow);
    hipHostFree(c->h_res);
    hipFree(c->d_rpsum); hipFree(c->d_ppkey); hipHostFree(c->h_radial);
    hipFree(c->d_rtab); hipHostFree(c->h_rtab); hipFree(c->d_ptab); hipHostFree(c->h_ptab); hipFree(c->d_wytab);
    if (c->s_copy) hipStreamDestroy(c->s_copy);
    if (c->s_post) hipStreamDestroy(c->s_post);
    delete c;
}

int ffl_create(int device, int width, int height, int n_frame_slots, int n_flow_slots, int max_batch,
               ffl_ctx **out) {
    if (!out) return set_err(nullptr, FFL_ERR_INVALID, "ffl_create: out is NULL");
    *out = nullptr;
    // upper bound: the kernels index a pair's 5 M planes (20 bytes per pixel) and a batch's pixels with 32-bit
    // offsets -> 20 * w * h must stay below 2^32 (about 214 Mpx; 5760x2880 is 16.6 Mpx)
    if (width < 16 || height < 16 || (long)width * height * 20 >= (1L << 32))
        return set_err(nullptr, FFL_ERR_INVALID, "ffl_create: unsupported frame size %dx%d (16x16 .. 20*w*h < 2^32)", width,
                       height);
    if (n_frame_slots < 2 || n_flow_slots < 1 || max_batch < 1 || max_batch > FFL_MAX_BATCH)
        return set_err(nullptr, FFL_ERR_INVALID, "ffl_create: bad slot/batch counts (%d, %d, %d)", n_frame_slots,
                       n_flow_slots, max_batch);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(nullptr, FFL_ERR_NO_DEVICE, "ffl_create: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev)
        return set_err(nullptr, FFL_ERR_INVALID, "ffl_create: device %d out of range (0..%d)", device, ndev - 1);
    ffl_ctx *c = new ffl_ctx();
    c->device = device;
    c->w = width;
    c->h = height;
    c->N = (size_t)width * height;
    c->n_fslots = n_frame_slots;
    c->n_slots = n_flow_slots;
    c->max_batch = max_batch;
    {
        std::lock_guard<std::mutex> g(g_opt_mu);
        c->opt = g_opts;
    }
    const int num_lanes = c->opt.lanes;  // fixed for the life of the context
    level_geometry(c);
    polyexp_prepare(&c->pc);
#define CCHK(call)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            int rc_ = set_err(nullptr, FFL_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
            ffl_destroy(c);                                                                              \
            (void)hipGetLastError(); /* the failure is reported HERE: do not leave it to poison a later call's check */ \
            return rc_;                                                                                  \
        }                                                                                                \
    } while (0)
    CCHK(hipSetDevice(device));
    // uploads and pass 2 are short and latency-critical (the host waits on pass 2): high priority, so
    // that they get their own hardware queues and are scheduled between a lane's queued kernels
    int prio_least = 0, prio_greatest = 0;
    CCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    CCHK(hipStreamCreateWithPriority(&c->s_copy, hipStreamNonBlocking, prio_greatest));
    CCHK(hipStreamCreateWithPriority(&c->s_post, hipStreamNonBlocking, prio_greatest));
    const size_t N = c->N;
    const int maxU = 2 * max_batch;
    CCHK(hipMalloc(&c->d_gray, (size_t)n_frame_slots * N + 16));  // +16: k_pyr_h fetches taps as aligned words
    CCHK(hipMalloc(&c->d_bgr, (size_t)n_frame_slots * N * 3));
    CCHK(hipHostMalloc(&c->h_stage_gray, (size_t)n_frame_slots * N, hipHostMallocDefault));
    CCHK(hipHostMalloc(&c->h_stage_bgr, (size_t)n_frame_slots * N * 3, hipHostMallocDefault));
    c->p1_blocks = ffl_pass1_blocks(width, height);
    c->lanes.resize(num_lanes);
    for (auto &L : c->lanes) {
        CCHK(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
        // st_aux (run-ahead / fork-join schedules only) are created on first use: HIP multiplexes streams
        // onto a few hardware queues, and idle extra streams make the latency-critical `post` / `copy`
        // streams share a queue with a compute lane (pass 2 then waits behind whole queued batches)
        CCHK(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
        CCHK(L.ring.create(FFL_EV_RING));
        size_t r_total = 0, i_total = 0, t_total = 0;
        for (int k = 0; k <= c->levels; k++) {
            CCHK(hipEventCreateWithFlags(&L.ev_R[k], hipEventDisableTiming));
            L.r_off[k] = r_total;
            L.i_off[k] = i_total;
            L.t_off[k] = t_total;
            t_total += ffl_pyr_tmp_floats(width, height, c->geom[k].lw) * maxU;
            r_total += (size_t)5 * c->geom[k].lw * c->geom[k].lh * maxU;
            i_total += (size_t)c->geom[k].lw * c->geom[k].lh * maxU;
        }
        CCHK(hipMalloc(&L.d_I, sizeof(float) * i_total));
        CCHK(hipMalloc(&L.d_T, sizeof(float) * t_total));
        CCHK(hipMalloc(&L.d_R, sizeof(float) * r_total));
        CCHK(hipMalloc(&L.d_M[0], sizeof(float) * 5 * N * max_batch));
        CCHK(hipMalloc(&L.d_M[1], sizeof(float) * 5 * N * max_batch));
        CCHK(hipMalloc(&L.d_flowA, sizeof(float) * 2 * N * max_batch));
        CCHK(hipMalloc(&L.d_flowB, sizeof(float) * 2 * N * max_batch));
        CCHK(hipMalloc(&L.d_pkey, sizeof(unsigned long long) * c->p1_blocks * max_batch));
        CCHK(hipMalloc(&L.d_psum, sizeof(double) * c->p1_blocks * max_batch));
        CCHK(hipMalloc(&L.d_tab, sizeof(BatchTab)));
        CCHK(hipHostMalloc(&L.h_tab, sizeof(BatchTab) * FFL_EV_RING, hipHostMallocDefault));
    }
    CCHK(hipMalloc(&c->d_flow, sizeof(float) * 2 * N * n_flow_slots));
    CCHK(hipHostMalloc(&c->h_res, sizeof(Pass1Result) * n_flow_slots, hipHostMallocMapped));
    CCHK(hipHostGetDevicePointer((void **)&c->d_res, c->h_res, 0));
    CCHK(hipMalloc(&c->d_rpsum, sizeof(double) * c->p1_blocks * FFL_MAXB));
    {
        std::vector<double> wy(2 * (size_t)height);
        for (int y = 0; y < height; y++) {
            wy[y] = (double)(height - y) / (double)height;
            wy[height + y] = (double)y / (double)height;
        }
        CCHK(hipMalloc(&c->d_wytab, sizeof(double) * wy.size()));
        CCHK(hipMemcpy(c->d_wytab, wy.data(), sizeof(double) * wy.size(), hipMemcpyHostToDevice));
    }
    CCHK(hipMalloc(&c->d_rtab, sizeof(RadialTab)));
    CCHK(hipHostMalloc(&c->h_rtab, sizeof(RadialTab), hipHostMallocDefault));
    CCHK(hipMalloc(&c->d_ptab, sizeof(BatchTab)));
    CCHK(hipHostMalloc(&c->h_ptab, sizeof(BatchTab), hipHostMallocDefault));
    CCHK(hipMalloc(&c->d_ppkey, sizeof(unsigned long long) * c->p1_blocks));
    CCHK(hipHostMalloc(&c->h_radial, sizeof(double) * FFL_MAXB, hipHostMallocMapped));
    CCHK(hipHostGetDevicePointer((void **)&c->d_radial, c->h_radial, 0));
    c->ev_uploaded.assign(n_frame_slots, EvRef{});
    c->ev_last_use.assign((size_t)n_frame_slots * num_lanes, EvRef{});  // references into the lanes' rings
    c->frame_valid.assign(n_frame_slots, 0);
    c->u_of_fslot.assign(n_frame_slots, -1);
    c->slot_mark.assign(n_flow_slots, 0);
    CCHK(c->up_ring.create(2 * FFL_EV_RING));
    CCHK(c->post_ring.create(FFL_EV_RING));
    c->ev_slot_done.assign(n_flow_slots, EvRef{});                        // set when a slot is queued
    c->slot_state.assign(n_flow_slots, 0);
    c->slot_pov.assign(n_flow_slots, 0);
#undef CCHK
    *out = c;
    return FFL_OK;
}

int ffl_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(nullptr, FFL_ERR_NO_DEVICE, "ffl_device_mem_info: no HIP device available");
    if (device < 0 || device >= ndev) return set_err(nullptr, FFL_ERR_INVALID, "ffl_device_mem_info: device %d out of range", device);
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess)
        return set_err(nullptr, FFL_ERR_HIP, "ffl_device_mem_info: hipMemGetInfo failed");
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return FFL_OK;
}

// What ffl_create(width, height, n_frame_slots, n_flow_slots, max_batch) allocates with the current "lanes" option: the
// same sums as the hipMalloc / hipHostMalloc calls above (small tables rounded up to 1 MiB in total).  No device needed.
int ffl_estimate_bytes(int width, int height, int n_frame_slots, int n_flow_slots, int max_batch, size_t *device_bytes,
                       size_t *pinned_bytes) {
    if (width < 16 || height < 16 || (long)width * height * 20 >= (1L << 32) || n_frame_slots < 2 || n_flow_slots < 1 ||
        max_batch < 1 || max_batch > FFL_MAX_BATCH)
        return set_err(nullptr, FFL_ERR_INVALID, "ffl_estimate_bytes: bad geometry / slot counts");
    ffl_ctx g;  // geometry only
    g.w = width;
    g.h = height;
    level_geometry(&g);
    const size_t N = (size_t)width * height, maxU = 2 * (size_t)max_batch;
    size_t lane = 0;
    for (int k = 0; k <= g.levels; k++) {
        const size_t n = (size_t)g.geom[k].lw * g.geom[k].lh;
        lane += sizeof(float) * maxU * (ffl_pyr_tmp_floats(width, height, g.geom[k].lw) + 5 * n + n);  // T, R, I
    }
    lane += sizeof(float) * N * max_batch * (5 + 5 + 2 + 2);                                            // M x 2, flow A / B
    lane += (size_t)ffl_pass1_blocks(width, height) * max_batch * 16;
    int num_lanes;
    {
        std::lock_guard<std::mutex> gl(g_opt_mu);
        num_lanes = g_opts.lanes;
    }
    size_t dev = (size_t)num_lanes * lane + (size_t)n_frame_slots * N * 4 + sizeof(float) * 2 * N * n_flow_slots + ((size_t)1 << 20);
    size_t pin = (size_t)n_frame_slots * N * 4 + sizeof(Pass1Result) * n_flow_slots + (size_t)num_lanes * sizeof(BatchTab) * FFL_EV_RING + ((size_t)1 << 20);
    if (device_bytes) *device_bytes = dev;
    if (pinned_bytes) *pinned_bytes = pin;
    return FFL_OK;
}

int ffl_num_levels(const ffl_ctx *c) { return c ? c->levels : -1; }

int ffl_level_size(const ffl_ctx *c, int level, int *out_wh) {
    if (!c || !out_wh || level < 0 || level > c->levels) return FFL_ERR_INVALID;
    out_wh[0] = c->geom[level].lw;
    out_wh[1] = c->geom[level].lh;
    return FFL_OK;
}

static bool in_host_buf(const ffl_ctx *c, const uint8_t *p, size_t bytes) {
    for (auto &hb : c->host_bufs)
        if (p >= hb.first && p + bytes <= hb.first + hb.second) return true;
    return false;
}

int ffl_host_alloc(ffl_ctx *c, size_t bytes, void **out) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (!out || bytes == 0) return set_err(c, FFL_ERR_INVALID, "ffl_host_alloc: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t *p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    c->host_bufs.push_back({p, bytes});
    *out = p;
    return FFL_OK;
}

int ffl_host_free(ffl_ctx *c, void *ptr) {
    if (!c) return FFL_ERR_INVALID;
    std::unique_lock<std::mutex> ul(c->up_mu);  // no upload out of the buffer starts while it is being freed
    CtxLock lk(c->mu);
    bool mine = false;
    for (auto &hb : c->host_bufs) mine |= hb.first == ptr;
    if (!mine) return set_err(c, FFL_ERR_INVALID, "ffl_host_free: not a buffer of this context");
    HIPCHK(c, hipSetDevice(c->device));
    lk.unlock();
    hipError_t e = hipStreamSynchronize(c->s_copy);  // no transfer may still be reading it (waited for without the lock)
    lk.lock();
    HIPCHK(c, e);
    for (size_t i = 0; i < c->host_bufs.size(); i++)
        if (c->host_bufs[i].first == ptr) {
            HIPCHK(c, hipHostFree(ptr));
            c->host_bufs.erase(c->host_bufs.begin() + i);
            return FFL_OK;
        }
    return set_err(c, FFL_ERR_INVALID, "ffl_host_free: not a buffer of this context");
}

// n frames into the consecutive frame slots first..first+n-1: host copies into pinned staging, then ONE
// H2D transfer (+ one BGR->gray launch) and ONE event for the whole run -- at 256x256 the per-frame
// runtime calls of a frame-at-a-time upload cost more than the copy itself.
int ffl_upload_frames(ffl_ctx *c, int first, int n, const uint8_t *const *frames, int width, int height, int channels,
                      ptrdiff_t stride_bytes) {
    if (!c) return FFL_ERR_INVALID;
    std::unique_lock<std::mutex> ul(c->up_mu);
    CtxLock lk(c->mu);
    if (!frames || n < 1 || first < 0 || first + n > c->n_fslots)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames: bad frame slot range %d..%d", first, first + n - 1);
    if (width != c->w || height != c->h)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames: frame is %dx%d, context is %dx%d", width, height, c->w, c->h);
    if (channels != 1 && channels != 3)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames: channels must be 1 (gray) or 3 (BGR), got %d", channels);
    if (stride_bytes < (ptrdiff_t)width * channels)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames: stride %td < row bytes %d", stride_bytes, width * channels);
    for (int i = 0; i < n; i++)
        if (!frames[i]) return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames: frame %d is NULL", i);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->N, row = (size_t)width * channels, fbytes = N * channels;
    uint8_t *stage0 = (channels == 1 ? c->h_stage_gray : c->h_stage_bgr) + (size_t)first * fbytes;
    // frames that sit back to back in one ffl_host_alloc buffer go to the device straight out of it
    bool direct = (size_t)stride_bytes == row && in_host_buf(c, frames[0], fbytes * n);
    for (int i = 1; direct && i < n; i++) direct = frames[i] == frames[0] + (size_t)i * fbytes;
    // Staged frames go to the device in pieces of >= 8 MiB as soon as they are in pinned memory, so the transfer of
    // the first frames runs while the host still copies the later ones (one 200 MB transfer issued after a 33-frame
    // BGR run had been staged left the device waiting for it); small frames still travel as one run.
    auto send = [&](int i0, int i1) -> int {  // frames i0 .. i1-1 of the run; called with the context lock held
        // the device copies of these slots may still be read by batches queued on any lane -- looked up HERE, right before
        // the transfer is queued: the lock was dropped for the staging copies, and a batch another thread queued meanwhile
        // must be ordered ahead of the transfer that overwrites its frame
        WaitOnce wait_copy(c->s_copy);
        for (int i = i0; i < i1; i++)
            for (size_t l = 0; l < c->lanes.size(); l++) HIPCHK(c, wait_copy(c->ev_last_use[(size_t)(first + i) * c->lanes.size() + l].get()));
        uint8_t *gray = c->d_gray + (size_t)(first + i0) * N;
        const uint8_t *src = (direct ? frames[0] : stage0) + (size_t)i0 * fbytes;
        const int m = i1 - i0;
        if (channels == 1) {
            HIPCHK(c, hipMemcpyAsync(gray, src, N * m, hipMemcpyHostToDevice, c->s_copy));
        } else {
            uint8_t *bgr = c->d_bgr + (size_t)(first + i0) * N * 3;
            HIPCHK(c, hipMemcpyAsync(bgr, src, N * 3 * m, hipMemcpyHostToDevice, c->s_copy));
            ProfScope ps(c, FFL_K_GRAY, c->s_copy);
            // k_gray counts pixels in 32 bits: runs of frames of at most 2^30 pixels per launch
            const int per = (int)((size_t)(1u << 30) / N) > 0 ? (int)((size_t)(1u << 30) / N) : 1;
            for (int i = 0; i < m; i += per) {
                const int k = m - i < per ? m - i : per;
                ffl_launch_gray(bgr + (size_t)i * N * 3, gray + (size_t)i * N, (int)(N * k), c->s_copy);
            }
        }
        return FFL_OK;
    };
    if (direct) {
        int rc = send(0, n);
        if (rc) return rc;
    } else {
        const int copy_threads = c->opt.copy_threads;
        int i = 0;
        while (i < n) {
            // one piece: consecutive frames worth >= 8 MiB (or the rest of the run), staged together -- the piece's rows
            // are shared by the copy threads -- and sent as one transfer
            int j = i;
            for (size_t bytes = 0; j < n && bytes < ((size_t)8 << 20); j++) bytes += fbytes;
            // the previous transfers out of these slots' staging areas must have left the host buffer; the waits and the
            // staging copy run WITHOUT the context lock (up_mu keeps other uploaders out of the staging areas and the pool)
            std::vector<hipEvent_t> prev;
            for (int k = i; k < j; k++) {
                hipEvent_t e = c->ev_uploaded[first + k].get();
                if (e && (prev.empty() || prev.back() != e)) prev.push_back(e);
            }
            lk.unlock();
            hipError_t pe = hipSuccess;
            for (auto e : prev)
                if (pe == hipSuccess) pe = hipEventSynchronize(e);
            if (pe == hipSuccess) c->pool.copy(stage0 + (size_t)i * fbytes, frames + i, j - i, stride_bytes, row, height, copy_threads);
            lk.lock();
            HIPCHK(c, pe);
            int rc = send(i, j);
            if (rc) return rc;
            i = j;
        }
    }
    HIPCHK(c, c->up_ring.settle_next());  // 32 upload calls old: over long ago
    EvRef ref{&c->up_ring, 0};
    HIPCHK(c, hipEventRecord(c->up_ring.take(&ref.ticket), c->s_copy));
    for (int i = 0; i < n; i++) {
        c->ev_uploaded[first + i] = ref;
        c->frame_valid[first + i] = 1;
    }
    return FFL_OK;
}

// Decoded frames -> gray frame slots through k_frontend (resize + crop + luma in one pass).  Frame by
// frame: tight copy into a pinned ring buffer, H2D, kernel -- the 3 * src_w * src_h byte transfer is the
// cost, so there is nothing to gain from batching the launches.
int ffl_upload_frames_raw(ffl_ctx *c, int first, int n, const uint8_t *const *frames, int sw, int sh,
                          ptrdiff_t stride_bytes, int rgb_order, int rw, int rh, int crop_x, int crop_y) {
    if (!c) return FFL_ERR_INVALID;
    std::unique_lock<std::mutex> ul(c->up_mu);
    CtxLock lk(c->mu);
    if (!frames || n < 1 || first < 0 || first + n > c->n_fslots)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames_raw: bad frame slot range %d..%d", first, first + n - 1);
    if (sw < 1 || sh < 1 || sw > 32768 || sh > 32768 || rw < 1 || rh < 1 || rw > 32768 || rh > 32768)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames_raw: unsupported source %dx%d / resize %dx%d", sw, sh, rw, rh);
    if (stride_bytes < (ptrdiff_t)sw * 3)
        return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames_raw: stride %td < row bytes %d", stride_bytes, sw * 3);
    if (crop_x < 0 || crop_y < 0 || crop_x + c->w > rw || crop_y + c->h > rh)
        return set_err(c, FFL_ERR_INVALID,
                       "ffl_upload_frames_raw: crop window (%d, %d) + %dx%d does not fit the %dx%d resized frame", crop_x,
                       crop_y, c->w, c->h, rw, rh);
    for (int i = 0; i < n; i++)
        if (!frames[i]) return set_err(c, FFL_ERR_INVALID, "ffl_upload_frames_raw: frame %d is NULL", i);
    HIPCHK(c, hipSetDevice(c->device));
    FrontParams fp;
    fp.sw = sw; fp.sh = sh;
    fp.stride = (size_t)sw * 3;
    fp.cx = crop_x; fp.cy = crop_y; fp.ow = c->w; fp.oh = c->h;
    fp.scale_x = 1. / ((double)rw / sw);
    fp.scale_y = 1. / ((double)rh / sh);
    fp.mode = (rw == sw && rh == sh) ? FFL_FRONT_IDENTITY
              : (sw == 2 * rw && sh == 2 * rh) ? FFL_FRONT_AREA2 : FFL_FRONT_GENERIC;
    fp.rgb = rgb_order != 0;
    const size_t fbytes = fp.stride * sh;
    for (int i = 0; i < n; i++) {
        const int fs = first + i;
        auto &rb = c->raw[c->raw_next++ % FFL_RAW_RING];
        if (rb.busy) {  // its previous frame has left both buffers (waited for without the context lock)
            lk.unlock();
            hipError_t be = hipEventSynchronize(rb.ev);
            lk.lock();
            HIPCHK(c, be);
        }
        if (rb.cap < fbytes) {
            hipFree(rb.d);
            hipHostFree(rb.h);
            rb.d = rb.h = nullptr;
            rb.cap = 0;
            HIPCHK(c, hipMalloc(&rb.d, fbytes));
            HIPCHK(c, hipHostMalloc(&rb.h, fbytes, hipHostMallocDefault));
            rb.cap = fbytes;
        }
        if (!rb.ev) HIPCHK(c, hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
        const uint8_t *data = frames[i];
        // a tightly packed frame in ffl_host_alloc memory goes to the device straight out of it
        const bool direct = (size_t)stride_bytes == fp.stride && in_host_buf(c, data, fbytes);
        if (!direct) {
            const int copy_threads = c->opt.copy_threads;
            lk.unlock();  // the staging copy runs without the context lock (up_mu protects the ring and the pool)
            c->pool.copy(rb.h, &data, 1, stride_bytes, fp.stride, sh, copy_threads);
            lk.lock();
        }
        for (size_t l = 0; l < c->lanes.size(); l++) {  // batches still reading the slot's previous frame
            hipEvent_t e = c->ev_last_use[(size_t)fs * c->lanes.size() + l].get();
            if (e) HIPCHK(c, hipStreamWaitEvent(c->s_copy, e, 0));
        }
        HIPCHK(c, hipMemcpyAsync(rb.d, direct ? data : rb.h, fbytes, hipMemcpyHostToDevice, c->s_copy));
        {
            ProfScope ps(c, FFL_K_FRONTEND, c->s_copy);
            ffl_launch_frontend(rb.d, c->d_gray + (size_t)fs * c->N, fp, c->s_copy);
        }
        HIPCHK(c, hipEventRecord(rb.ev, c->s_copy));
        rb.busy = true;
    }
    HIPCHK(c, c->up_ring.settle_next());
    EvRef ev{&c->up_ring, 0};
    HIPCHK(c, hipEventRecord(c->up_ring.take(&ev.ticket), c->s_copy));
    for (int i = 0; i < n; i++) {
        // a gray upload into this slot may still be in flight out of the slot's own staging area; the new
        // handle is later on the same stream, so waiting on it covers that transfer as well
        c->ev_uploaded[first + i] = ev;
        c->frame_valid[first + i] = 1;
    }
    return FFL_OK;
}

int ffl_upload_frame(ffl_ctx *c, int fslot, const uint8_t *data, int width, int height, int channels,
                     ptrdiff_t stride_bytes) {
    return ffl_upload_frames(c, fslot, 1, &data, width, height, channels, stride_bytes);
}

int ffl_download_frame(ffl_ctx *c, int fslot, uint8_t *dst) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (!dst || fslot < 0 || fslot >= c->n_fslots)
        return set_err(c, FFL_ERR_INVALID, "ffl_download_frame: bad frame slot %d", fslot);
    if (!c->frame_valid[fslot]) return set_err(c, FFL_ERR_STATE, "ffl_download_frame: frame slot %d was never uploaded", fslot);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, c->d_gray + (size_t)fslot * c->N, c->N, hipMemcpyDeviceToHost, c->s_copy));
    HIPCHK(c, hipStreamSynchronize(c->s_copy));
    return FFL_OK;
}

struct DebugCapture {
    int level, iter;
    float *I0, *I1, *R0, *R1, *M, *flow;
};

// The launches of one batch (frame expansion, the level loop, pass 1) on the lane's stream.  Everything that
// differs between two batches of the same shape is read by the kernels from the lane's device table, so this
// sequence can be captured into a hipGraph (cap == nullptr, no timing events, serial schedule).
static int enqueue_batch(ffl_ctx *c, ffl_ctx::Lane &L, const BatchTab &T, int n, int nU, int pov_mode, const DebugCapture *cap) {
    hipStream_t st = L.st;
    const size_t N = c->N;
    const UTab *ut = &L.d_tab->ut;
    const PairTab *pt = &L.d_tab->pt;
    // frame-only expansion of level k (level image + PolyExp of the nU unique frames) on stream s
    auto expand_level = [&](int k, hipStream_t s) {
        const LevelGeom &g = c->geom[k];
        const size_t plane = (size_t)g.lw * g.lh;
        {
            ProfScope ps(c, FFL_K_PYRAMID, s);
            ffl_launch_pyr_level(c->d_gray, N, ut, nU, c->w, c->h, g.lw, g.lh, g.gk, L.d_T + L.t_off[k],
                                 ffl_pyr_tmp_floats(c->w, c->h, g.lw), L.d_I + L.i_off[k], plane, s);
        }
        {
            ProfScope ps(c, FFL_K_POLYEXP, s);
            ffl_launch_polyexp(L.d_I + L.i_off[k], plane, L.d_R + L.r_off[k], 5 * plane, plane, nU, g.lw, g.lh, c->pc, s);
        }
    };
    // Frame-only expansion schedule (ffl_set_option "run_ahead"):
    //   0  serial, on the lane's stream, level by level (also used for debug capture)
    //   2  fork/join: the 4 levels expand concurrently on side streams (their small grids and serial
    //      LDS phases leave most of the device idle when run one after the other), and the flow chain
    //      starts only after all of them -- it is never co-scheduled with anything
    //   1  run-ahead: one side stream, the chain waits per level, so coarse-level flow kernels overlap
    //      the finer levels' expansion (fastest with one lane, but stretches those launches)
    const int mode = cap ? 0 : c->opt.run_ahead;
    if (mode) {
        for (int k = 0; k < (mode == 2 ? 4 : 1); k++)
            if (!L.st_aux[k]) HIPCHK(c, hipStreamCreateWithFlags(&L.st_aux[k], hipStreamNonBlocking));
        HIPCHK(c, hipEventRecord(L.ev_fork, st));  // after the uploads and after the lane's previous batch
        for (int k = c->levels; k >= 0; k--) {
            hipStream_t sa = L.st_aux[mode == 2 ? k : 0];
            if (mode == 2 || k == c->levels) HIPCHK(c, hipStreamWaitEvent(sa, L.ev_fork, 0));
            expand_level(k, sa);
            HIPCHK(c, hipEventRecord(L.ev_R[k], sa));
        }
        if (mode == 2)
            for (int k = c->levels; k >= 0; k--) HIPCHK(c, hipStreamWaitEvent(st, L.ev_R[k], 0));
    }
    const bool run_ahead = mode == 1;
    bool expanded = mode != 0;
    if (mode == 0 && c->opt.merge_expand && c->levels + 1 <= FFL_MAX_JOBS) {
        // serial schedule, merged form: the frame-only work of ALL levels up front in three launches
        // (pyramid phase A + B, PolyExp) instead of ten small ones whose ramps and tails leave the device idle
        PyrJob pj[FFL_MAX_JOBS];
        PolyJob qj[FFL_MAX_JOBS];
        int nj = 0;
        for (int k = c->levels; k >= 0; k--, nj++) {
            const LevelGeom &g = c->geom[k];
            const size_t plane = (size_t)g.lw * g.lh;
            memset(&pj[nj], 0, sizeof(PyrJob));
            pj[nj].lw = g.lw;
            pj[nj].lh = g.lh;
            pj[nj].gk = g.gk;
            pj[nj].tmp = L.d_T + L.t_off[k];
            pj[nj].tmp_stride = ffl_pyr_tmp_floats(c->w, c->h, g.lw);
            pj[nj].I = L.d_I + L.i_off[k];
            pj[nj].I_stride = plane;
            memset(&qj[nj], 0, sizeof(PolyJob));
            qj[nj].I = L.d_I + L.i_off[k];
            qj[nj].I_stride = plane;
            qj[nj].R = L.d_R + L.r_off[k];
            qj[nj].R_stride = 5 * plane;
            qj[nj].plane = plane;
            qj[nj].w = g.lw;
            qj[nj].h = g.lh;
        }
        {
            ProfScope ps(c, FFL_K_PYRAMID, st);
            if (!ffl_launch_pyr_multi(c->d_gray, N, ut, nU, c->w, c->h, pj, nj, c->opt, st))
                for (int k = c->levels; k >= 0; k--) {  // a level outside the merged kinds: per-level kernels
                    const LevelGeom &g = c->geom[k];
                    ffl_launch_pyr_level(c->d_gray, N, ut, nU, c->w, c->h, g.lw, g.lh, g.gk, L.d_T + L.t_off[k],
                                         ffl_pyr_tmp_floats(c->w, c->h, g.lw), L.d_I + L.i_off[k], (size_t)g.lw * g.lh, st);
                }
        }
        {
            ProfScope ps(c, FFL_K_POLYEXP, st);
            ffl_launch_polyexp_multi(qj, nj, nU, c->pc, st);
        }
        expanded = true;
    }

    int pw = 0, ph = 0;
    for (int k = c->levels; k >= 0; k--) {
        const LevelGeom &g = c->geom[k];
        const int lw = g.lw, lh = g.lh;
        const size_t plane = (size_t)lw * lh;
        const size_t I_stride = plane, R_stride = 5 * plane, M_stride = 5 * plane;
        float *Rk = L.d_R + L.r_off[k];
        // the coarsest level starts from zero flow; nothing reads that field but UpdateMatrices (which is told
        // so) and the debug capture, so it is only materialised for the latter
        if (pw == 0 && cap)
            for (int i = 0; i < n; i++) HIPCHK(c, hipMemsetAsync(T.pt.flow[k][i], 0, sizeof(float) * 2 * plane, st));
        if (run_ahead) HIPCHK(c, hipStreamWaitEvent(st, L.ev_R[k], 0));
        else if (!expanded) expand_level(k, st);
        int mi = 0;
        // fuse_first: the level's initial UpdateMatrices (and flow upsample) run inside the first blur+solve
        // launch; the debug capture wants the initial flow and M in memory, so it keeps the separate launch
        const bool fuse_first = c->opt.fuse_first > 0 && !cap && (long)((lw + 63) / 64) * ((lh + 15) / 16) * n >= c->opt.fuse_first;
        if (!fuse_first) {
            ProfScope ps(c, FFL_K_UPDATE_MATRICES, st);
            // the x2 upsample of the coarser level's flow (K3) is fused into this launch
            ffl_launch_update_matrices(Rk, R_stride, plane, pt, k, n, L.d_M[mi], M_stride, lw, lh, pw, ph, pw == 0, cap != nullptr, c->opt, st);
        }
        bool captured = false;
        auto capture = [&]() -> int {
            HIPCHK(c, hipStreamSynchronize(st));
            if (cap->I0) HIPCHK(c, hipMemcpy(cap->I0, L.d_I + L.i_off[k] + (size_t)T.pt.u0[0] * I_stride, sizeof(float) * plane, hipMemcpyDeviceToHost));
            if (cap->I1) HIPCHK(c, hipMemcpy(cap->I1, L.d_I + L.i_off[k] + (size_t)T.pt.u1[0] * I_stride, sizeof(float) * plane, hipMemcpyDeviceToHost));
            if (cap->R0) HIPCHK(c, hipMemcpy(cap->R0, Rk + (size_t)T.pt.u0[0] * R_stride, sizeof(float) * 5 * plane, hipMemcpyDeviceToHost));
            if (cap->R1) HIPCHK(c, hipMemcpy(cap->R1, Rk + (size_t)T.pt.u1[0] * R_stride, sizeof(float) * 5 * plane, hipMemcpyDeviceToHost));
            if (cap->M) HIPCHK(c, hipMemcpy(cap->M, L.d_M[mi], sizeof(float) * 5 * plane, hipMemcpyDeviceToHost));
            if (cap->flow) HIPCHK(c, hipMemcpy(cap->flow, T.pt.flow[k][0], sizeof(float) * 2 * plane, hipMemcpyDeviceToHost));
            captured = true;
            return FFL_OK;
        };
        for (int it = 0; it < 3; it++) {
            if (cap && cap->level == k && cap->iter == it) {
                int rc = capture();
                if (rc) return rc;
            }
            const int update = it < 2;
            {
                ProfScope ps(c, FFL_K_BLUR_SOLVE, st, it > 0 && !cap);  // the three iterations are queued back to back
                if (it == 0 && fuse_first)
                    ffl_launch_blur_solve_first(L.d_M[mi ^ 1], M_stride, Rk, R_stride, plane, pt, k, n, lw, lh, pw, ph, c->opt, st);
                else
                    ffl_launch_blur_solve(L.d_M[mi], L.d_M[mi ^ 1], M_stride, Rk, R_stride, plane, pt, k, n, lw, lh,
                                          update, cap != nullptr, c->opt, st);
            }
            if (update) mi ^= 1;
        }
        if (cap && cap->level == k && !captured) {
            int rc = capture();
            if (rc) return rc;
        }
        pw = lw;
        ph = lh;
    }
    // pass 1 on the finished level-0 flows; records are stored straight into mapped pinned memory
    {
        ProfScope ps(c, FFL_K_PASS1, st);
        ffl_launch_pass1(pt, n, c->w, c->h, pov_mode, L.d_pkey, L.d_psum, st);
    }
    return FFL_OK;
}

// One batch of pairs through the 4-scale Farneback schedule + pass-1 reductions, on compute lane `li`.
static int run_batch(ffl_ctx *c, int li, int n, const int *f0, const int *f1, const int *slots, int pov_mode,
                     const DebugCapture *cap) {
    ffl_ctx::Lane &L = c->lanes[li];
    hipStream_t st = L.st;
    const size_t N = c->N;
    // the batch's table: a pinned ring entry (its previous batch, FFL_EV_RING batches ago, must have consumed it)
    const unsigned e = (unsigned)(L.ring.next % FFL_EV_RING);
    HIPCHK(c, L.ring.settle_next());  // also frees the table entry: its batch, FFL_EV_RING batches ago, has consumed it
    BatchTab &T = L.h_tab[e];
    int nU = 0;
    auto uidx = [&](int fs) {  // O(1) through the context's scratch map (a linear search cost 65 k compares per 256-pair batch)
        int &u = c->u_of_fslot[fs];
        if (u < 0) {
            T.ut.fslot[nU] = fs;
            u = nU++;
        }
        return u;
    };
    for (int i = 0; i < n; i++) {
        T.pt.u0[i] = uidx(f0[i]);
        T.pt.u1[i] = uidx(f1[i]);
        T.pt.res[i] = c->d_res + slots[i];
    }
    // per level: the pair's flow field -- the lane's two ping-pong buffers, coarsest level in A; level 0 is the slot
    for (int k = c->levels; k >= 0; k--) {
        const size_t plane = (size_t)c->geom[k].lw * c->geom[k].lh;
        float *buf = ((c->levels - k) & 1) ? L.d_flowB : L.d_flowA;
        for (int i = 0; i < n; i++)
            T.pt.flow[k][i] = (k == 0) ? c->d_flow + (size_t)slots[i] * 2 * N : buf + (size_t)i * 2 * plane;
    }
    for (int k = c->levels + 1; k < FFL_MAX_LEVELS; k++)
        for (int i = 0; i < n; i++) T.pt.flow[k][i] = nullptr;
    for (int i = 0; i < nU; i++) c->u_of_fslot[T.ut.fslot[i]] = -1;  // the scratch map goes back to "empty"
    WaitOnce wait(st);
    for (int i = 0; i < nU; i++) HIPCHK(c, wait(c->ev_uploaded[T.ut.fslot[i]].get()));
    // a flow slot being recycled may still be read by the batch (other lane) or pass 2 that used it last
    for (int i = 0; i < n; i++)
        if (c->slot_state[slots[i]]) HIPCHK(c, wait(c->ev_slot_done[slots[i]].get()));
    // stream order puts this copy behind the lane's previous batch, which reads the same device table
    HIPCHK(c, hipMemcpyAsync(L.d_tab, &T, sizeof(BatchTab), hipMemcpyHostToDevice, st));

    const bool use_graph = c->opt.use_graph && !cap && c->prof_mask == 0 && c->opt.run_ahead == 0;
    if (use_graph) {
        ffl_ctx::Lane::GraphEntry *ge = nullptr;
        for (auto &g : L.graphs)
            if (g.n == n && g.nU == nU && g.pov == pov_mode && g.epoch == c->opt_epoch) ge = &g;
        if (!ge && c->graph_bad_epoch != c->opt_epoch) {
            ffl_ctx::Lane::GraphEntry g = {n, nU, pov_mode, c->opt_epoch, nullptr, nullptr};
            hipError_t fail = hipSuccess;
            fail = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            bool ok = fail == hipSuccess;
            if (ok) {
                const int rc = enqueue_batch(c, L, T, n, nU, pov_mode, nullptr);
                fail = hipStreamEndCapture(st, &g.graph);  // always: the stream must leave capture mode
                if (fail == hipSuccess && (rc != FFL_OK || !g.graph)) fail = hipErrorStreamCaptureInvalidated;
                if (fail == hipSuccess) fail = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
                ok = fail == hipSuccess;
            }
            if (ok) {
                if (L.graphs.size() >= 16) {  // bounded cache: callers that vary the batch shape a lot re-capture
                    // a replay may still be queued: graph resources are released once the lane's stream has drained
                    HIPCHK(c, hipStreamSynchronize(st));
                    hipGraphExecDestroy(L.graphs.front().exec);
                    hipGraphDestroy(L.graphs.front().graph);
                    L.graphs.erase(L.graphs.begin());
                }
                L.graphs.push_back(g);
                ge = &L.graphs.back();
                c->graph_captured++;
            } else {
                // nothing of the failed capture is kept (it would leak once per batch), the sticky error is cleared, and
                // this context launches eagerly until the option set changes -- the batch itself is not lost
                if (g.exec) hipGraphExecDestroy(g.exec);
                if (g.graph) hipGraphDestroy(g.graph);
                (void)hipGetLastError();
                c->graph_bad_epoch = c->opt_epoch;
                // never silent: counted (ffl_graph_stats, bench.py `config.graphs`) and said once per context on stderr
                c->graph_failed++;
                if (!c->graph_fail_reported) {
                    c->graph_fail_reported = true;
                    fprintf(stderr, "libffl_hip: hipGraph capture of a %d-pair batch failed (%s); this context launches its batches "
                                    "one kernel at a time until its options change (results are unaffected)\n", n, hipGetErrorString(fail));
                }
            }
        }
        if (ge) {
            HIPCHK(c, hipGraphLaunch(ge->exec, st));
            c->graph_replayed++;
        } else {
            int rc = enqueue_batch(c, L, T, n, nU, pov_mode, nullptr);
            if (rc) return rc;
        }
    } else {
        int rc = enqueue_batch(c, L, T, n, nU, pov_mode, cap);
        if (rc) return rc;
    }
    // ONE event per batch: it marks the slots' results as ready, the frames' last use and the lane's
    // work buffers as free (a ring, so handles held by older slots only ever point to later work)
    EvRef ev{&L.ring, 0};
    HIPCHK(c, hipEventRecord(L.ring.take(&ev.ticket), st));
    for (int i = 0; i < nU; i++) c->ev_last_use[(size_t)T.ut.fslot[i] * c->lanes.size() + li] = ev;
    for (int i = 0; i < n; i++) {
        c->ev_slot_done[slots[i]] = ev;
        c->slot_state[slots[i]] = 1;
        c->slot_pov[slots[i]] = (char)(pov_mode != 0);
    }
    HIPCHK(c, hipGetLastError());
    return FFL_OK;
}

static int check_pairs(ffl_ctx *c, int n, const int *f0, const int *f1, const int *slots) {
    if (n < 1 || n > c->max_batch) return set_err(c, FFL_ERR_INVALID, "batch of %d pairs, context allows 1..%d", n, c->max_batch);
    if (!f0 || !f1 || !slots) return set_err(c, FFL_ERR_INVALID, "NULL slot table");
    for (int i = 0; i < n; i++) {
        if (f0[i] < 0 || f0[i] >= c->n_fslots || f1[i] < 0 || f1[i] >= c->n_fslots)
            return set_err(c, FFL_ERR_INVALID, "pair %d: frame slot out of range", i);
        if (!c->frame_valid[f0[i]] || !c->frame_valid[f1[i]])
            return set_err(c, FFL_ERR_STATE, "pair %d: frame slot was never uploaded", i);
        if (slots[i] < 0 || slots[i] >= c->n_slots) return set_err(c, FFL_ERR_INVALID, "pair %d: flow slot out of range", i);
    }
    int dup = -1;  // O(n) through the context's scratch marks
    for (int i = 0; i < n; i++) {
        if (c->slot_mark[slots[i]] && dup < 0) dup = slots[i];
        c->slot_mark[slots[i]] = 1;
    }
    for (int i = 0; i < n; i++) c->slot_mark[slots[i]] = 0;
    if (dup >= 0) return set_err(c, FFL_ERR_INVALID, "flow slot %d used twice in one batch", dup);
    return FFL_OK;
}

int ffl_flow_pairs(ffl_ctx *c, int n, const int *fslot0, const int *fslot1, const int *flow_slots, int pov_mode) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    int rc = check_pairs(c, n, fslot0, fslot1, flow_slots);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int li = (int)(c->next_lane++ % c->lanes.size());
    return run_batch(c, li, n, fslot0, fslot1, flow_slots, pov_mode, nullptr);
}

int ffl_debug_pair(ffl_ctx *c, int f0, int f1, int level, int iter, float *I0, float *I1, float *R0, float *R1,
                   float *M, float *flow) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    int slot = 0;
    int rc = check_pairs(c, 1, &f0, &f1, &slot);
    if (rc) return rc;
    if (level < 0 || level > c->levels || iter < 0 || iter > 3) return set_err(c, FFL_ERR_INVALID, "bad level/iter");
    HIPCHK(c, hipSetDevice(c->device));
    DebugCapture cap = {level, iter, I0, I1, R0, R1, M, flow};
    rc = run_batch(c, 0, 1, &f0, &f1, &slot, 0, &cap);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->lanes[0].st));
    return FFL_OK;
}

int ffl_pass1_result(ffl_ctx *c, int slot, float cut_threshold, int32_t *x, int32_t *y, float *div_val, float *mean_mag,
                     int *cut) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (slot < 0 || slot >= c->n_slots) return set_err(c, FFL_ERR_INVALID, "flow slot %d out of range", slot);
    if (!c->slot_state[slot]) return set_err(c, FFL_ERR_STATE, "flow slot %d holds no result", slot);
    HIPCHK(c, hipSetDevice(c->device));
    {
        // wait without the lock: the event handle is a ring entry that is only ever re-recorded for LATER work
        // of the same lane, so waiting on it after another thread queued more batches is still sufficient
        hipEvent_t ev = c->ev_slot_done[slot].get();
        lk.unlock();
        hipError_t e = ev ? hipEventSynchronize(ev) : hipSuccess;
        lk.lock();
        HIPCHK(c, e);
    }
    const Pass1Result &r = c->h_res[slot];
    float mm = (float)(r.mag_sum / ((double)c->w * (double)c->h));
    if (x) *x = r.x;
    if (y) *y = r.y;
    if (div_val) *div_val = r.div_val;
    if (mean_mag) *mean_mag = mm;
    if (cut) *cut = mm > cut_threshold ? 1 : 0;
    return FFL_OK;
}

int ffl_pass1_results(ffl_ctx *c, int n, const int *slots, float cut_threshold, int32_t *x, int32_t *y, float *div_val,
                      float *mean_mag, int *cut) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (n < 0 || (n > 0 && !slots)) return set_err(c, FFL_ERR_INVALID, "ffl_pass1_results: bad arguments");
    // one lock, one wait per DISTINCT event (the slots of a batch share theirs), then the records: 256 one-slot calls
    // cost 0.3 ms of host time per batch at the 256x256 operating point
    hipEvent_t evs[16];
    int ne = 0;
    for (int i = 0; i < n; i++) {
        const int slot = slots[i];
        if (slot < 0 || slot >= c->n_slots) return set_err(c, FFL_ERR_INVALID, "flow slot %d out of range", slot);
        if (!c->slot_state[slot]) return set_err(c, FFL_ERR_STATE, "flow slot %d holds no result", slot);
        hipEvent_t e = c->ev_slot_done[slot].get();
        bool seen = !e;
        for (int k = 0; k < ne && !seen; k++) seen = evs[k] == e;
        if (!seen) {
            if (ne == 16) {  // more distinct events than a call normally names: wait for what has been collected, go on
                lk.unlock();
                hipError_t we = hipSuccess;
                for (int k = 0; k < ne && we == hipSuccess; k++) we = hipEventSynchronize(evs[k]);
                lk.lock();
                HIPCHK(c, we);
                ne = 0;
            }
            evs[ne++] = e;
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    {
        lk.unlock();  // the wait does not hold up uploads / submissions of other threads
        hipError_t we = hipSuccess;
        for (int k = 0; k < ne && we == hipSuccess; k++) we = hipEventSynchronize(evs[k]);
        lk.lock();
        HIPCHK(c, we);
    }
    const double npx = (double)c->w * (double)c->h;
    for (int i = 0; i < n; i++) {
        const Pass1Result &r = c->h_res[slots[i]];
        const float mm = (float)(r.mag_sum / npx);
        if (x) x[i] = r.x;
        if (y) y[i] = r.y;
        if (div_val) div_val[i] = r.div_val;
        if (mean_mag) mean_mag[i] = mm;
        if (cut) cut[i] = mm > cut_threshold ? 1 : 0;
    }
    return FFL_OK;
}

int ffl_radial(ffl_ctx *c, int n, const int *slots, const double *cx, const double *cy, const int *is_cut, int pov_mode,
               double *out) {
    if (!c) return FFL_ERR_INVALID;
    std::unique_lock<std::mutex> pl(c->post_mu);  // one pass-2 call at a time owns stream `post`, h_rtab and h_radial
    CtxLock lk(c->mu);
    if (n < 1 || n > FFL_MAXB || !slots || !cx || !cy || !out) return set_err(c, FFL_ERR_INVALID, "ffl_radial: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    RadialTab &rt = *c->h_rtab;  // the previous call waited for s_post, so the pinned copy is free
    WaitOnce wait_post(c->s_post);
    int m = 0;
    int map[FFL_MAXB];
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || slots[i] >= c->n_slots) return set_err(c, FFL_ERR_INVALID, "flow slot %d out of range", slots[i]);
        if (!c->slot_state[slots[i]]) return set_err(c, FFL_ERR_STATE, "flow slot %d holds no flow", slots[i]);
        if (is_cut && is_cut[i]) {  // FF:766-767: a cut returns 0.0 without looking at the flow
            out[i] = 0.0;
            continue;
        }
        HIPCHK(c, wait_post(c->ev_slot_done[slots[i]].get()));
        rt.flow[m] = c->d_flow + (size_t)slots[i] * 2 * c->N;
        rt.cx[m] = cx[i];
        rt.cy[m] = cy[i];
        map[m++] = i;
    }
    if (m == 0) return FFL_OK;
    hipStream_t st = c->s_post;
    {
        HIPCHK(c, hipMemcpyAsync(c->d_rtab, &rt, sizeof(RadialTab), hipMemcpyHostToDevice, st));
        ProfScope ps(c, FFL_K_RADIAL, st);
        ffl_launch_radial(c->d_rtab, m, c->w, c->h, pov_mode, c->d_wytab, c->d_rpsum, c->d_radial, st);
    }
    {
        // the slots' "last use" now includes this pass 2: the wait below runs without the context lock, so another thread
        // may queue a batch that recycles one of these slots meanwhile -- it must run behind the kernel that reads them
        HIPCHK(c, c->post_ring.settle_next());
        EvRef ev{&c->post_ring, 0};
        HIPCHK(c, hipEventRecord(c->post_ring.take(&ev.ticket), st));
        for (int j = 0; j < m; j++) c->ev_slot_done[slots[map[j]]] = ev;
    }
    lk.unlock();  // the wait (for the batches the slots come from, then pass 2) does not hold up uploads / submissions
    hipError_t se = hipStreamSynchronize(st);  // k_radial_final stored into the mapped pinned buffer
    lk.lock();
    HIPCHK(c, se);
    HIPCHK(c, hipGetLastError());
    for (int j = 0; j < m; j++) out[map[j]] = c->h_radial[j];
    return FFL_OK;
}

int ffl_download_flow(ffl_ctx *c, int slot, float *dst) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (slot < 0 || slot >= c->n_slots || !dst) return set_err(c, FFL_ERR_INVALID, "ffl_download_flow: bad arguments");
    if (!c->slot_state[slot]) return set_err(c, FFL_ERR_STATE, "flow slot %d holds no flow", slot);
    HIPCHK(c, hipSetDevice(c->device));
    {
        hipEvent_t ev = c->ev_slot_done[slot].get();
        lk.unlock();
        hipError_t e = ev ? hipEventSynchronize(ev) : hipSuccess;
        lk.lock();
        HIPCHK(c, e);
    }
    HIPCHK(c, hipMemcpy(dst, c->d_flow + (size_t)slot * 2 * c->N, sizeof(float) * 2 * c->N, hipMemcpyDeviceToHost));
    return FFL_OK;
}

int ffl_upload_flow(ffl_ctx *c, int slot, const float *src, int pov_mode) {
    if (!c) return FFL_ERR_INVALID;
    std::unique_lock<std::mutex> pl(c->post_mu);
    CtxLock lk(c->mu);
    if (slot < 0 || slot >= c->n_slots || !src) return set_err(c, FFL_ERR_INVALID, "ffl_upload_flow: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->s_post;
    {
        hipEvent_t ev = c->slot_state[slot] ? c->ev_slot_done[slot].get() : nullptr;
        lk.unlock();
        hipError_t e = ev ? hipEventSynchronize(ev) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        lk.lock();
        HIPCHK(c, e);
    }
    HIPCHK(c, hipMemcpy(c->d_flow + (size_t)slot * 2 * c->N, src, sizeof(float) * 2 * c->N, hipMemcpyHostToDevice));
    c->h_ptab->pt.flow[0][0] = c->d_flow + (size_t)slot * 2 * c->N;  // the stream was drained above: the pinned copy is free
    c->h_ptab->pt.res[0] = c->d_res + slot;
    HIPCHK(c, hipMemcpyAsync(c->d_ptab, c->h_ptab, sizeof(BatchTab), hipMemcpyHostToDevice, st));
    {
        ProfScope ps(c, FFL_K_PASS1, st);
        ffl_launch_pass1(&c->d_ptab->pt, 1, c->w, c->h, pov_mode, c->d_ppkey, c->d_rpsum, st);
    }
    HIPCHK(c, c->post_ring.settle_next());
    c->ev_slot_done[slot] = EvRef{&c->post_ring, 0};
    HIPCHK(c, hipEventRecord(c->post_ring.take(&c->ev_slot_done[slot].ticket), st));
    c->slot_state[slot] = 1;
    c->slot_pov[slot] = (char)(pov_mode != 0);
    HIPCHK(c, hipGetLastError());
    return FFL_OK;
}

int ffl_submit_pair(ffl_ctx *c, int slot, const uint8_t *prev, const uint8_t *next, int width, int height, int channels,
                    ptrdiff_t stride_bytes, int pov_mode) {
    if (!c) return FFL_ERR_INVALID;
    // no lock around the three calls (each takes its own; holding `mu` here would invert the up_mu -> mu order): calls on
    // distinct slots from several threads interleave safely, n_fslots / n_slots are fixed at creation
    if (slot < 0 || 2 * slot + 1 >= c->n_fslots || slot >= c->n_slots) {
        CtxLock lk(c->mu);
        return set_err(c, FFL_ERR_INVALID, "ffl_submit_pair: slot %d needs frame slots %d,%d and a flow slot", slot, 2 * slot, 2 * slot + 1);
    }
    int rc = ffl_upload_frame(c, 2 * slot, prev, width, height, channels, stride_bytes);
    if (rc) return rc;
    rc = ffl_upload_frame(c, 2 * slot + 1, next, width, height, channels, stride_bytes);
    if (rc) return rc;
    int f0 = 2 * slot, f1 = 2 * slot + 1;
    return ffl_flow_pairs(c, 1, &f0, &f1, &slot, pov_mode);
}

int ffl_sync(ffl_ctx *c) {
    if (!c) return FFL_ERR_INVALID;
    // up_mu / post_mu: an upload or pass-2 call of another thread that is under way finishes queueing first, so the latest
    // events of streams `copy` and `post` stand for everything those calls put there
    std::unique_lock<std::mutex> ul(c->up_mu);
    std::unique_lock<std::mutex> pl(c->post_mu);
    CtxLock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    // Wait on EVENTS, never on the lane streams: without the lock another thread may be capturing a new batch shape on a
    // lane's stream, and hipStreamSynchronize on a capturing stream fails and invalidates the capture.  Each handle is a
    // ring entry that is only ever re-recorded for LATER work of its stream, so waiting on it stays sufficient.
    std::vector<hipEvent_t> evs;
    if (hipEvent_t e = ev_latest(c->up_ring).get()) evs.push_back(e);
    for (auto &rb : c->raw)
        if (rb.busy && rb.ev) evs.push_back(rb.ev);
    for (auto &L : c->lanes)
        if (hipEvent_t e = ev_latest(L.ring).get()) evs.push_back(e);
    if (hipEvent_t e = ev_latest(c->post_ring).get()) evs.push_back(e);
    lk.unlock();
    hipError_t e = hipSuccess;
    for (auto ev : evs)
        if (e == hipSuccess) e = hipEventSynchronize(ev);
    lk.lock();
    HIPCHK(c, e);
    return FFL_OK;
}

// one knob of an option set; `live`: the set belongs to an existing context (its lane count is fixed)
static int set_option_impl(FflOptions &o, const char *name, int value, bool live) {
    if (!strcmp(name, "blur_tile_h")) {  // fixed: the box-sum order is anchored to blocks of 16 rows
        return value == 16 ? FFL_OK : FFL_ERR_INVALID;
    }
    if (!strcmp(name, "fuse_first")) {  // minimum tiles x pairs of a level for the folded first launch; 0: never
        if (value < 0) return FFL_ERR_INVALID;
        o.fuse_first = value;
        return FFL_OK;
    }
    if (!strcmp(name, "merge_expand")) {  // 1 (default): merged frame-expansion launches, 0: one set per level
        o.merge_expand = value != 0;
        return FFL_OK;
    }
    if (!strcmp(name, "blur_rows")) {  // tiles a k_blur_solve workgroup walks down: 0 automatic, 1..64
        if (value < 0 || value > 64) return FFL_ERR_INVALID;
        o.blur_rows = value;
        return FFL_OK;
    }
    if (!strcmp(name, "blur_min_wgs")) {  // automatic strip length: the longest strips that still give this many workgroups
        if (value < 1) return FFL_ERR_INVALID;
        o.blur_min_wgs = value;
        return FFL_OK;
    }
    if (!strcmp(name, "tile_order")) {  // 0 pair-major, 1 tile-major (ffl_tile_coord)
        if (value < 0 || value > 1) return FFL_ERR_INVALID;
        o.tile_order = value;
        return FFL_OK;
    }
    if (!strcmp(name, "pyr_coarse")) {  // 1 (default): one-pass kernel for the two coarse pyramid levels, 0: H + V kernel pairs
        o.pyr_coarse = value != 0;
        return FFL_OK;
    }
    if (!strcmp(name, "copy_threads")) {  // host threads sharing a staging copy of >= 1 MiB (1 = the caller alone)
        if (value < 1 || value > 16) return FFL_ERR_INVALID;
        o.copy_threads = value;
        return FFL_OK;
    }
    if (!strcmp(name, "graph")) {  // 1 (default): replay a batch's launches from a captured hipGraph, 0: launch eagerly
        o.use_graph = value != 0;
        return FFL_OK;
    }
    if (!strcmp(name, "lanes")) {  // compute lanes: a property of the context's buffers, fixed at ffl_create
        if (value < 1 || value > 4) return FFL_ERR_INVALID;
        if (live) return value == o.lanes ? FFL_OK : FFL_ERR_STATE;
        o.lanes = value;
        return FFL_OK;
    }
    if (!strcmp(name, "run_ahead")) {  // frame-only expansion schedule: 0 serial, 1 run-ahead, 2 fork/join
        if (value < 0 || value > 2) return FFL_ERR_INVALID;
        o.run_ahead = value;
        return FFL_OK;
    }
    return FFL_ERR_INVALID;
}

static int get_option_impl(const FflOptions &o, const char *name, int *value) {
    struct { const char *n; int v; } tab[] = {
        {"blur_tile_h", 16}, {"fuse_first", o.fuse_first}, {"merge_expand", o.merge_expand}, {"blur_rows", o.blur_rows},
        {"blur_min_wgs", o.blur_min_wgs}, {"tile_order", o.tile_order}, {"pyr_coarse", o.pyr_coarse},
        {"copy_threads", o.copy_threads}, {"graph", o.use_graph}, {"lanes", o.lanes}, {"run_ahead", o.run_ahead}};
    for (auto &t : tab)
        if (!strcmp(name, t.n)) {
            *value = t.v;
            return FFL_OK;
        }
    return FFL_ERR_INVALID;
}

int ffl_set_option(const char *name, int value) {
    if (!name) return FFL_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_opt_mu);
    return set_option_impl(g_opts, name, value, false);
}

int ffl_ctx_set_option(ffl_ctx *c, const char *name, int value) {
    if (!c || !name) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    const int rc = set_option_impl(c->opt, name, value, true);
    if (rc == FFL_OK) c->opt_epoch++;  // only an option that was actually applied invalidates this context's captured graphs
    else if (rc == FFL_ERR_STATE) set_err(c, rc, "ffl_ctx_set_option: \"lanes\" is fixed once the context exists (%d)", c->opt.lanes);
    else set_err(c, rc, "ffl_ctx_set_option: unknown option or value out of range: %s = %d", name, value);
    return rc;
}

int ffl_ctx_get_option(ffl_ctx *c, const char *name, int *value) {
    if (!name || !value) return FFL_ERR_INVALID;
    if (!c) {
        std::lock_guard<std::mutex> g(g_opt_mu);
        return get_option_impl(g_opts, name, value);
    }
    CtxLock lk(c->mu);
    return get_option_impl(c->opt, name, value);
}

int ffl_graph_stats(ffl_ctx *c, int *captured, int *replayed, int *capture_failures) {
    if (!c) return FFL_ERR_INVALID;
    CtxLock lk(c->mu);
    if (captured) *captured = c->graph_captured;
    if (replayed) *replayed = c->graph_replayed;
    if (capture_failures) *capture_failures = c->graph_failed;
    return FFL_OK;
}

int ffl_profile_enable(ffl_ctx *c, unsigned class_mask) {
    if (!c) return FFL_ERR_INVALID;
    ffl_sync(c);  // before the context lock: ffl_sync takes up_mu / post_mu first (lock order)
    CtxLock lk(c->mu);
    prof_collect(c);
    c->prof_mask = class_mask;
    return FFL_OK;
}

int ffl_profile_read(ffl_ctx *c, int k, int *launches, double *total_ms) {
    if (!c || k < 0 || k >= FFL_K_COUNT) return FFL_ERR_INVALID;
    int rc = ffl_sync(c);  // before the context lock (lock order up_mu -> post_mu -> mu)
    if (rc) return rc;
    CtxLock lk(c->mu);
    prof_collect(c);  // launches another thread queued since the sync are waited for here, under the lock (measurement hook)
    if (launches) *launches = c->prof_launches[k];
    if (total_ms) *total_ms = c->prof_ms[k];
    c->prof_launches[k] = 0;
    c->prof_ms[k] = 0;
    return FFL_OK;
}

}  // extern "C"
