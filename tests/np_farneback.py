"""tests/np_farneback.py -- TEST INFRASTRUCTURE: a second, independent restatement of SURVEY.md Appendix A.

Written from the appendix alone (whole-array numpy, cumulative-sum box filter, `numpy.linalg.inv`), NOT from
oracle/farneback_oracle.c: the two share the specification and nothing else, so agreement between them
(tests/test_oracle_farneback.py::test_second_restatement_agrees) catches a slip in either transcription -- an index,
a sign, a channel order, a border rule -- that the analytic known-answer tests do not reach.  It cannot catch an error
in Appendix A itself: cv2 is not importable here and the reference holds no Farneback fixtures ("parity unpinned").
Float32 element-wise operations are written in the appendix's order, so most stages agree to the last bit; the box
filter sums in a different order (double), hence a tolerance on the final flow.
"""
import numpy as np

f32 = np.float32


def gaussian_kernel(n, sigma):                                   # A.2
    if sigma <= 0 and n == 3:
        return np.array([0.25, 0.5, 0.25], f32)
    if sigma <= 0:
        sigma = 0.3 * ((n - 1) / 2 - 1) + 0.8
    x = np.arange(n, dtype=np.float64) - (n - 1) / 2
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(f32)


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def gaussian_blur(img, n, sigma):                                # A.2: separable, f32, REFLECT_101, rows then columns
    k = gaussian_kernel(n, sigma)
    r = n // 2

    def along(a, axis):
        m = a.shape[axis]
        idx = np.arange(m)
        acc = np.take(a, idx, axis) * k[r]
        for j in range(1, r + 1):
            acc = acc + k[r + j] * (np.take(a, _reflect101(idx - j, m), axis) + np.take(a, _reflect101(idx + j, m), axis))
        return acc.astype(f32)

    return along(along(img.astype(f32), 1), 0)


def _resize_axis(src_n, dst_n):
    s = src_n / dst_n
    d = np.arange(dst_n)
    f = ((d + 0.5) * s - 0.5).astype(f32)
    i = np.floor(f).astype(np.int64)
    f = (f - i.astype(f32)).astype(f32)
    lo = i < 0
    i[lo], f[lo] = 0, 0
    hi = i >= src_n - 1
    i[hi], f[hi] = src_n - 1, 0
    return i, np.minimum(i + 1, src_n - 1), f


def resize_linear(a, w, h):                                      # A.2: horizontal lerp, then vertical, f32 weights
    a = a.astype(f32)
    x0, x1, fx = _resize_axis(a.shape[1], w)
    y0, y1, fy = _resize_axis(a.shape[0], h)
    shape_x = (1, w) + (1,) * (a.ndim - 2)
    shape_y = (h, 1) + (1,) * (a.ndim - 2)
    fx, fy = fx.reshape(shape_x), fy.reshape(shape_y)
    rows = (a[:, x0] * (f32(1) - fx) + a[:, x1] * fx).astype(f32)
    return (rows[y0] * (f32(1) - fy) + rows[y1] * fy).astype(f32)


def polyexp(I, n=5, sigma=1.2):                                  # A.3
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2 * sigma * sigma)).astype(f32)
    g = (g.astype(np.float64) * (1.0 / g.astype(np.float64).sum())).astype(f32)   # f32 values, 1/sum in double
    xg = (x.astype(f32) * g).astype(f32)
    xxg = (x.astype(f32) * x.astype(f32) * g).astype(f32)
    gd = g.astype(np.float64)
    G = np.zeros((6, 6))
    for yy in range(-n, n + 1):
        for xx in range(-n, n + 1):
            gg = gd[yy + n] * gd[xx + n]
            G[0, 0] += gg
            G[1, 1] += gg * xx * xx
            G[3, 3] += gg * xx ** 4
            G[5, 5] += gg * xx * xx * yy * yy
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    ig11, ig03, ig33, ig55 = inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]
    h, w = I.shape
    I = I.astype(f32)
    ys = np.arange(h)
    row0 = I * g[n]
    row1 = np.zeros_like(I)
    row2 = np.zeros_like(I)
    for k in range(1, n + 1):
        a, b = I[np.maximum(ys - k, 0)], I[np.minimum(ys + k, h - 1)]
        row0 = (row0 + g[n + k] * (a + b)).astype(f32)
        row1 = (row1 + xg[n + k] * (b - a)).astype(f32)
        row2 = (row2 + xxg[n + k] * (a + b)).astype(f32)
    xs = np.arange(w)
    D = np.float64
    b1, b3, b5 = (row0 * g[n]).astype(D), (row1 * g[n]).astype(D), (row2 * g[n]).astype(D)
    b2 = np.zeros((h, w)); b4 = np.zeros((h, w)); b6 = np.zeros((h, w))
    for k in range(1, n + 1):
        p, m = np.minimum(xs + k, w - 1), np.maximum(xs - k, 0)
        s0 = (row0[:, p] + row0[:, m]).astype(f32)
        b1 = b1 + s0.astype(D) * D(g[n + k])
        b4 = b4 + s0.astype(D) * D(xxg[n + k])
        b2 = b2 + ((row0[:, p] - row0[:, m]).astype(f32) * xg[n + k]).astype(f32).astype(D)
        b3 = b3 + ((row1[:, p] + row1[:, m]).astype(f32) * g[n + k]).astype(f32).astype(D)
        b6 = b6 + ((row1[:, p] - row1[:, m]).astype(f32) * xg[n + k]).astype(f32).astype(D)
        b5 = b5 + ((row2[:, p] + row2[:, m]).astype(f32) * g[n + k]).astype(f32).astype(D)
    return np.stack([(b3 * ig11).astype(f32), (b2 * ig11).astype(f32), (b1 * ig03 + b5 * ig33).astype(f32),
                     (b1 * ig03 + b4 * ig33).astype(f32), (b6 * ig55).astype(f32)], axis=-1)


BORDER = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], f32)


def update_matrices(R0, R1, flow):                               # A.4
    h, w = flow.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w]
    dx, dy = flow[..., 0].astype(f32), flow[..., 1].astype(f32)
    fx, fy = (xs.astype(f32) + dx).astype(f32), (ys.astype(f32) + dy).astype(f32)
    x1, y1 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    fx, fy = (fx - x1.astype(f32)).astype(f32), (fy - y1.astype(f32)).astype(f32)
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = np.clip(x1, 0, w - 2), np.clip(y1, 0, h - 2)
    one = f32(1)
    a00, a01, a10, a11 = ((one - fx) * (one - fy)).astype(f32), (fx * (one - fy)).astype(f32), ((one - fx) * fy).astype(f32), (fx * fy).astype(f32)
    r = []
    for c in range(5):
        P = R1[..., c]
        v = (a00 * P[yc, xc]).astype(f32)
        v = (v + a01 * P[yc, xc + 1]).astype(f32)
        v = (v + a10 * P[yc + 1, xc]).astype(f32)
        v = (v + a11 * P[yc + 1, xc + 1]).astype(f32)
        r.append(v)
    r2 = np.where(inside, r[0], f32(0)).astype(f32)
    r3 = np.where(inside, r[1], f32(0)).astype(f32)
    r4 = np.where(inside, ((R0[..., 2] + r[2]) * f32(0.5)).astype(f32), R0[..., 2])
    r5 = np.where(inside, ((R0[..., 3] + r[3]) * f32(0.5)).astype(f32), R0[..., 3])
    r6 = np.where(inside, ((R0[..., 4] + r[4]) * f32(0.25)).astype(f32), (R0[..., 4] * f32(0.5)).astype(f32))
    r2 = ((R0[..., 0] - r2) * f32(0.5)).astype(f32)
    r3 = ((R0[..., 1] - r3) * f32(0.5)).astype(f32)
    r2 = (r2 + ((r4 * dy).astype(f32) + (r6 * dx).astype(f32)).astype(f32)).astype(f32)
    r3 = (r3 + ((r6 * dy).astype(f32) + (r5 * dx).astype(f32)).astype(f32)).astype(f32)

    # (x<5 ? b[x] : 1) * (x>=w-5 ? b[w-x-1] : 1) * (y<5 ? ...) * (y>=h-5 ? ...), left to right
    sx_lo = np.where(xs < 5, BORDER[np.minimum(xs, 4)], one).astype(f32)
    sx_hi = np.where(xs >= w - 5, BORDER[np.clip(w - xs - 1, 0, 4)], one).astype(f32)
    sy_lo = np.where(ys < 5, BORDER[np.minimum(ys, 4)], one).astype(f32)
    sy_hi = np.where(ys >= h - 5, BORDER[np.clip(h - ys - 1, 0, 4)], one).astype(f32)
    s = (((sx_lo * sx_hi).astype(f32) * sy_lo).astype(f32) * sy_hi).astype(f32)
    r2, r3, r4, r5, r6 = [(v * s).astype(f32) for v in (r2, r3, r4, r5, r6)]
    return np.stack([((r4 * r4).astype(f32) + (r6 * r6).astype(f32)).astype(f32),
                     ((r4 + r5).astype(f32) * r6).astype(f32),
                     ((r5 * r5).astype(f32) + (r6 * r6).astype(f32)).astype(f32),
                     ((r4 * r2).astype(f32) + (r6 * r3).astype(f32)).astype(f32),
                     ((r6 * r2).astype(f32) + (r5 * r3).astype(f32)).astype(f32)], axis=-1)


def blur_solve(M, m=7):                                          # A.5: 15x15 box (replicate), double, then the 2x2 solve
    P = np.pad(M.astype(np.float64), ((m, m), (m, m), (0, 0)), mode="edge")
    C = np.zeros((P.shape[0] + 1, P.shape[1] + 1, 5))
    C[1:, 1:] = P.cumsum(0).cumsum(1)
    n = 2 * m + 1
    B = C[n:, n:] - C[:-n, n:] - C[n:, :-n] + C[:-n, :-n]
    sc = 1.0 / (n * n)
    g11, g12, g22, h1, h2 = [B[..., c] * sc for c in range(5)]
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    return np.stack([((g11 * h2 - g12 * h1) * idet).astype(f32), ((g22 * h1 - g12 * h2) * idet).astype(f32)], axis=-1)


def cv_round(v):
    return int(np.rint(v))  # half to even


def farneback(prev, nxt, pyr_scale=0.5, levels=3, winsize=15, iters=3, poly_n=5, poly_sigma=1.2):   # A.1
    rows, cols = prev.shape
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if cols * scale < 32 or rows * scale < 32:
            break
        k += 1
    levels = k
    flow = None
    for k in range(levels, -1, -1):
        scale = pyr_scale ** k
        sigma = (1.0 / scale - 1) * 0.5
        smooth = max(cv_round(sigma * 5) | 1, 3)
        w, h = cv_round(cols * scale), cv_round(rows * scale)
        if flow is None:
            flow = np.zeros((h, w, 2), f32)
        else:
            flow = (resize_linear(flow, w, h) * f32(1.0 / pyr_scale)).astype(f32)
        R = []
        for img in (prev, nxt):
            fimg = gaussian_blur(img.astype(f32), smooth, sigma)
            R.append(polyexp(resize_linear(fimg, w, h), poly_n, poly_sigma))
        M = update_matrices(R[0], R[1], flow)
        for it in range(iters):
            flow = blur_solve(M, winsize // 2)
            if it < iters - 1:
                M = update_matrices(R[0], R[1], flow)
    return flow
