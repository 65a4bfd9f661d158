"""Seeded synthetic (frame, render stack, warp stack) workloads of the shapes in BASELINE.json / SURVEY.md 8(d).

Input generation only (numpy, host): the reference gets its render stack from OpenGL
(Thirdparty/Localization/rendering.hpp:530-630) and its warp stack from cv::cuda::warpPerspective
(Thirdparty/Localization/image.cpp:115-128); neither exists here, so benches and tests feed the scoring path
with images of the same format: uint8, render background 255 (rendering.hpp:533), warp border 0.
"""
import numpy as np

# Examples/Monocular/ETH_small.yaml:8-11,23-24 (960x540 calibration)
ETH_FX, ETH_FY, ETH_CX, ETH_CY, ETH_W, ETH_H = 435.04593205, 435.04593205, 475.55781765, 274.7487729, 960, 540


def _box_blur(a, radius):
    k = 2 * radius + 1
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (radius + 1, radius)
        c = np.cumsum(np.pad(a, pad, mode="wrap"), axis=axis)
        a = (np.take(c, np.arange(k, k + a.shape[axis]), axis=axis) - np.take(c, np.arange(0, a.shape[axis]), axis=axis)) / k
    return a


def scene(width, height, seed=1234):
    """Smooth natural-image-like scene: box-blurred (radius 8) Gaussian field, mean 128, sigma 40, uint8."""
    g = _box_blur(np.random.default_rng(seed).standard_normal((height, width)), 8)
    g = (g - g.mean()) / g.std()
    return np.clip(np.rint(128 + 40 * g), 0, 255).astype(np.uint8)


def camera_frame(scene_u8, seed=1235, noise_sigma=10.0):
    n = np.random.default_rng(seed).normal(0.0, noise_sigma, scene_u8.shape)
    return np.clip(np.rint(scene_u8.astype(np.float64) + n), 0, 255).astype(np.uint8)


def grid_counts(n):
    """(nx, ny, nz) for a stack of n = nx*ny*nz images: the most cubic factorisation (27 -> 3,3,3; 64 -> 4,4,4)."""
    best = None
    for nz in range(1, n + 1):
        if n % nz:
            continue
        for ny in range(1, n // nz + 1):
            if (n // nz) % ny:
                continue
            nx = n // nz // ny
            cand = (max(nx, ny, nz) - min(nx, ny, nz), nz, ny, nx)
            if best is None or cand < best:
                best = cand
    return best[3], best[2], best[1]


def render_stack(scene_u8, counts, gamma=0.7, shift_px=4, zoom_step=0.02, bottom_up=False):
    """[S][H][W] renders: the scene through a monotone gamma LUT (modality gap), shifted by
    shift_px*(sx-cx, sy-cy) pixels and zoomed by 1+zoom_step*(sz-cz) (nearest); uncovered pixels = 255.
    s = (sz*ny + sy)*nx + sx.  Centre cell = the unshifted render (planted optimum)."""
    nx, ny, nz = counts
    h, w = scene_u8.shape
    lut = np.clip(np.rint(255.0 * (np.arange(256) / 255.0) ** gamma), 0, 255).astype(np.uint8)
    base = lut[scene_u8]
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.empty((nx * ny * nz, h, w), np.uint8)
    for sz in range(nz):
        z = 1.0 + zoom_step * (sz - nz // 2)
        for sy in range(ny):
            for sx in range(nx):
                dx, dy = shift_px * (sx - nx // 2), shift_px * (sy - ny // 2)
                u = np.rint((xx - w / 2.0) / z + w / 2.0 - dx).astype(np.int64)
                v = np.rint((yy - h / 2.0) / z + h / 2.0 - dy).astype(np.int64)
                ok = (u >= 0) & (u < w) & (v >= 0) & (v < h)
                img = np.full((h, w), 255, np.uint8)
                img[ok] = base[v[ok], u[ok]]
                out[(sz * ny + sy) * nx + sx] = img[::-1] if bottom_up else img
    return out


def intrinsics(width, height):
    sx, sy = width / ETH_W, height / ETH_H
    return np.array([[ETH_FX * sx, 0, ETH_CX * sx], [0, ETH_FY * sy, ETH_CY * sy], [0, 0, 1.0]])


def warp_homographies(K, counts, steps):
    """K*Rz*Ry*Rx*K^-1 per warp cell, Thirdparty/Localization/image.cpp:76-107, including its integer
    division in the start angle (-(n-1)/2*step with int n).  Returns [Wn][3][3], w = (wz*ny+wy)*nx+wx."""
    nx, ny, nz = counts
    sx, sy, sz = steps
    Kinv = np.linalg.inv(K)
    out = np.empty((nx * ny * nz, 3, 3))
    for i in range(nz):
        tz = -((nz - 1) // 2) * sz + i * sz
        Rz = np.array([[np.cos(tz), -np.sin(tz), 0], [np.sin(tz), np.cos(tz), 0], [0, 0, 1]])
        for j in range(ny):
            ty = -((ny - 1) // 2) * sy + j * sy
            Ry = np.array([[np.cos(ty), 0, np.sin(ty)], [0, 1, 0], [-np.sin(ty), 0, np.cos(ty)]])
            for k in range(nx):
                tx = -((nx - 1) // 2) * sx + k * sx
                Rx = np.array([[1, 0, 0], [0, np.cos(tx), -np.sin(tx)], [0, np.sin(tx), np.cos(tx)]])
                out[(i * ny + j) * nx + k] = K @ (Rz @ Ry @ Rx) @ Kinv
    return out


def warp_perspective(img, M):
    """dst(x, y) = bilinear src at M^-1 (x, y, 1), constant border 0 -- the semantics of
    cv::cuda::warpPerspective as called at image.cpp:123 (forward matrix, INTER_LINEAR, BORDER_CONSTANT)."""
    h, w = img.shape
    Mi = np.linalg.inv(M)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = Mi[2, 0] * xx + Mi[2, 1] * yy + Mi[2, 2]
    u = (Mi[0, 0] * xx + Mi[0, 1] * yy + Mi[0, 2]) / d
    v = (Mi[1, 0] * xx + Mi[1, 1] * yy + Mi[1, 2]) / d
    x0, y0 = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
    fx, fy = u - x0, v - y0
    src = np.pad(img.astype(np.float64), 1)  # zero border

    def at(yi, xi):
        ok = (xi >= -1) & (xi <= w) & (yi >= -1) & (yi <= h)
        return np.where(ok, src[np.clip(yi, -1, h) + 1, np.clip(xi, -1, w) + 1], 0.0)

    val = ((1 - fx) * (1 - fy) * at(y0, x0) + fx * (1 - fy) * at(y0, x0 + 1)
           + (1 - fx) * fy * at(y0 + 1, x0) + fx * fy * at(y0 + 1, x0 + 1))
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


def warp_stack(frame_u8, counts, steps=(0.02, 0.02, 0.05)):
    h, w = frame_u8.shape
    Ms = warp_homographies(intrinsics(w, h), counts, steps)
    return np.stack([warp_perspective(frame_u8, M) for M in Ms])


def workload(width, height, S, Wn, seed=1234, bottom_up=True):
    """-> dict(frame, render_stack [S,H,W], warp_stack [Wn,H,W], s_counts, w_counts, planted (w*S+s of the centre))."""
    sc, wc = grid_counts(S), grid_counts(Wn)
    B = scene(width, height, seed)
    F = camera_frame(B, seed + 1)
    rs = render_stack(B, sc, bottom_up=bottom_up)
    ws = warp_stack(F, wc)
    s_c = ((sc[2] // 2) * sc[1] + sc[1] // 2) * sc[0] + sc[0] // 2
    w_c = (((wc[2] - 1) // 2) * wc[1] + (wc[1] - 1) // 2) * wc[0] + (wc[0] - 1) // 2  # identity warp cell (image.cpp:77)
    return {"frame": F, "render_stack": rs, "warp_stack": ws, "s_counts": sc, "w_counts": wc,
            "planted": w_c * S + s_c, "bottom_up": bottom_up}


def uniform_pair(width, height, seed=1234):
    """Independent uniform-random pair of SURVEY.md 8(c): rng(seed).integers twice."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (height, width), dtype=np.uint8)
    b = rng.integers(0, 256, (height, width), dtype=np.uint8)
    return a, b
