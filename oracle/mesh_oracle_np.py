"""numpy (fp32) restatement of the textured-mesh render-stack producer -- TEST INFRASTRUCTURE ONLY.

Restates Rendering<1>::renderToTextureOnGPU (Thirdparty/Localization/rendering.hpp:530-630), the shaders
ShadingWithTexture.* (luma 0.299/0.587/0.114 of the texture sample) and the texture state of loadBMP_custom
(texture.cpp:31-96: GL_REPEAT, GL_LINEAR, GL_LINEAR_MIPMAP_LINEAR, glGenerateMipmap) with the OpenGL 3.3 specification's
rules in fp32: pixel centres at +0.5, top-left fill rule, back-face culling (front = counter-clockwise), perspective-
correct attributes (u/w, v/w, 1/w as planes over the window, one reciprocal per sample point), visibility by 24-bit depth
with GL_LESS (among equal depths the triangle drawn first), isotropic LOD from per-pixel uv differences, 2x2-box mip levels
rounded to RGB8, and clipping of
triangles against the near plane in clip space (GL clips primitives to the view volume before the perspective divide:
implied by glEnable(GL_DEPTH_TEST) / glDrawArrays(GL_TRIANGLES), rendering.hpp:294-300,619): Sutherland-Hodgman on
z_clip >= -w_clip, new corners interpolated from the inside corner towards the outside one.
"PARITY UNPINNED": an OpenGL driver rasterises in fixed point and may approximate the LOD; nothing to compare against.
Slow (python loop over triangles): small test meshes only.
"""
import numpy as np

f32 = np.float32


def mip_luma(rgb):
    """-> list of float32 luma levels (row 0 = v 0), mip chain by 2x2 box on RGB8 with rounding, like nmi_texture_create."""
    cur = np.ascontiguousarray(rgb, np.uint8)
    levels = []
    while True:
        c = cur.astype(f32) / f32(255.0)
        levels.append((f32(0.299) * c[..., 0] + f32(0.587) * c[..., 1]) + f32(0.114) * c[..., 2])
        h, w = cur.shape[:2]
        if (w == 1 and h == 1) or len(levels) == 16:
            break
        nw, nh = max(1, w // 2), max(1, h // 2)
        x0 = np.minimum(2 * np.arange(nw), w - 1)
        x1 = np.minimum(2 * np.arange(nw) + 1, w - 1)
        y0 = np.minimum(2 * np.arange(nh), h - 1)
        y1 = np.minimum(2 * np.arange(nh) + 1, h - 1)
        s = (cur[np.ix_(y0, x0)].astype(np.int32) + cur[np.ix_(y0, x1)] + cur[np.ix_(y1, x0)] + cur[np.ix_(y1, x1)])
        cur = ((s + 2) // 4).astype(np.uint8)
    return levels


def _bilinear(level, u, v):
    h, w = level.shape
    x = u * f32(w) - f32(0.5)
    y = v * f32(h) - f32(0.5)
    xf, yf = np.floor(x), np.floor(y)
    fx, fy = x - xf, y - yf
    i0 = np.mod(xf.astype(np.int64), w)
    j0 = np.mod(yf.astype(np.int64), h)
    i1 = np.where(i0 + 1 == w, 0, i0 + 1)
    j1 = np.where(j0 + 1 == h, 0, j0 + 1)
    t00, t10, t01, t11 = level[j0, i0], level[j0, i1], level[j1, i0], level[j1, i1]
    a = t00 + (t10 - t00) * fx
    b = t01 + (t11 - t01) * fx
    return a + (b - a) * fy


def render_mesh(xyz, uv, levels, mvp_colmajor, width, height):
    """One view -> uint8 [H, W], bottom-up rows, background 255."""
    m = np.asarray(mvp_colmajor, f32)
    P = np.asarray(xyz, f32).reshape(-1, 3, 3)
    T = np.asarray(uv, f32).reshape(-1, 3, 2)
    zbuf = np.full((height, width), 0xFFFFFFFFFF, np.uint64)   # depth << 8 | grey; empty: a depth no fragment reaches, grey 255
    for tri in range(P.shape[0]):
        x, y, z = P[tri, :, 0], P[tri, :, 1], P[tri, :, 2]
        cx = (m[0] * x + m[4] * y) + (m[8] * z + m[12])
        cy = (m[1] * x + m[5] * y) + (m[9] * z + m[13])
        cz = (m[2] * x + m[6] * y) + (m[10] * z + m[14])
        cw = (m[3] * x + m[7] * y) + (m[11] * z + m[15])
        tu, tv = T[tri, :, 0], T[tri, :, 1]
        for sub in _clip_near(cx, cy, cz, cw, tu, tv):
            _raster(zbuf, levels, width, height, *sub)
    return (zbuf & np.uint64(0xFF)).astype(np.uint8)


def _clip_near(cx, cy, cz, cw, tu, tv):
    """Near-plane clipping of one triangle in clip space -> list of 0, 1 or 2 triangles, each (cx, cy, cz, cw, tu, tv) of 3."""
    d = cz + cw
    inside = d >= 0
    n_in = int(inside.sum())
    if n_in == 0:
        return []
    if n_in == 3:
        return [(cx, cy, cz, cw, tu, tv)]
    poly = []
    for k in range(3):
        b = (k + 1) % 3
        if inside[k]:
            poly.append((cx[k], cy[k], cz[k], cw[k], tu[k], tv[k]))
        if inside[k] != inside[b]:
            i, o = (k, b) if inside[k] else (b, k)   # from the inside corner towards the outside one
            t = d[i] / (d[i] - d[o])
            w = cw[i] + (cw[o] - cw[i]) * t
            poly.append((cx[i] + (cx[o] - cx[i]) * t, cy[i] + (cy[o] - cy[i]) * t, -w, w, tu[i] + (tu[o] - tu[i]) * t,
                         tv[i] + (tv[o] - tv[i]) * t))
    out = []
    for a_, b_, c_ in ((0, 1, 2), (0, 2, 3))[:len(poly) - 2]:
        out.append(tuple(np.array([poly[a_][j], poly[b_][j], poly[c_][j]], f32) for j in range(6)))
    return out


def _raster(zbuf, levels, width, height, cx, cy, cz, cw, tu, tv):
    tw, th = f32(levels[0].shape[1]), f32(levels[0].shape[0])
    nlev = len(levels)
    if True:
        if not (cw > 0).all():
            return
        if ((cx < -cw).all() or (cx > cw).all() or (cy < -cw).all() or (cy > cw).all() or (cz < -cw).all() or (cz > cw).all()):
            return
        xw = (cx / cw * f32(0.5) + f32(0.5)) * f32(width)
        yw = (cy / cw * f32(0.5) + f32(0.5)) * f32(height)
        zw = cz / cw * f32(0.5) + f32(0.5)
        iw = f32(1.0) / cw
        area = (xw[1] - xw[0]) * (yw[2] - yw[0]) - (xw[2] - xw[0]) * (yw[1] - yw[0])
        if not area > 0:
            return
        x_lo = max(0, int(np.ceil(xw.min() - f32(0.5))))
        x_hi = min(width - 1, int(np.floor(xw.max() - f32(0.5))))
        y_lo = max(0, int(np.ceil(yw.min() - f32(0.5))))
        y_hi = min(height - 1, int(np.floor(yw.max() - f32(0.5))))
        if x_lo > x_hi or y_lo > y_hi:
            return
        inv_area = f32(1.0) / area
        ex = np.array([xw[(k + 2) % 3] - xw[(k + 1) % 3] for k in range(3)], f32)
        ey = np.array([yw[(k + 2) % 3] - yw[(k + 1) % 3] for k in range(3)], f32)
        own = [bool(ey[k] < 0 or (ey[k] == 0 and ex[k] < 0)) for k in range(3)]
        yy, xx = np.mgrid[y_lo:y_hi + 1, x_lo:x_hi + 1]
        fxp = xx.astype(f32) + f32(0.5)
        fyp = yy.astype(f32) + f32(0.5)

        def weights(px, py):
            return [(ex[k] * (py - yw[(k + 1) % 3]) - ey[k] * (px - xw[(k + 1) % 3])) * inv_area for k in range(3)]

        # coverage and depth: barycentric weights per pixel (tri_cover, nmi_mesh.hip)
        b = weights(fxp, fyp)
        zz = (b[0] * zw[0] + b[1] * zw[1]) + b[2] * zw[2]
        inside = np.ones(fxp.shape, bool)
        for k in range(3):
            inside &= (b[k] > 0) | ((b[k] == 0) & own[k])
        inside &= (zz >= 0) & (zz <= 1)
        if not inside.any():
            return
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            depth = np.minimum((zz * f32(16777215.0) + f32(0.5)).astype(np.uint32), np.uint32(0xFFFFFF))
        # Visibility: GL_LESS keeps the fragment drawn first among equal depths, and triangles arrive here in draw order, so
        # a fragment wins only with a strictly smaller depth (the key depth << 40 | triangle << 10 | ... of nmi_mesh.hip).
        sub = zbuf[y_lo:y_hi + 1, x_lo:x_hi + 1]
        win = inside & (depth.astype(np.uint64) < (sub >> np.uint64(8)))
        if not win.any():
            return
        # attributes: u/w, v/w and 1/w as planes about the centre of the box's first pixel (tri_planes / shade_pixel)
        xr, yr = f32(x_lo) + f32(0.5), f32(y_lo) + f32(0.5)
        b0 = weights(xr, yr)
        bx = [(-ey[k]) * inv_area for k in range(3)]
        by = [ex[k] * inv_area for k in range(3)]
        sc = [tu[k] * iw[k] for k in range(3)]
        rc = [tv[k] * iw[k] for k in range(3)]

        def plane(g):
            return ((b0[0] * g[0] + b0[1] * g[1]) + b0[2] * g[2], (bx[0] * g[0] + bx[1] * g[1]) + bx[2] * g[2],
                    (by[0] * g[0] + by[1] * g[1]) + by[2] * g[2])

        (s0, sx, sy), (r0, rx, ry), (q0, qx, qy) = plane(sc), plane(rc), plane(iw)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            dx, dy = fxp - xr, fyp - yr
            S = (s0 + sx * dx) + sy * dy
            R = (r0 + rx * dx) + ry * dy
            Q = (q0 + qx * dx) + qy * dy
            iq, iqx, iqy = f32(1.0) / Q, f32(1.0) / (Q + qx), f32(1.0) / (Q + qy)
            u, v = S * iq, R * iq
            ux, vx = (S + sx) * iqx, (R + rx) * iqx
            uy, vy = (S + sy) * iqy, (R + ry) * iqy
            dudx, dvdx, dudy, dvdy = (ux - u) * tw, (vx - v) * th, (uy - u) * tw, (vy - v) * th
            rho2 = np.maximum(dudx * dudx + dvdx * dvdx, dudy * dudy + dvdy * dvdy)
            lam = f32(0.5) * np.log2(rho2).astype(f32)      # log2 of the longer footprint axis = half the log2 of its square
            luma = _bilinear(levels[0], u, v)
            mini = lam > 0
            if mini.any():
                lc = np.minimum(lam, f32(nlev - 1))
                l0 = np.floor(lc).astype(np.int64)
                fr = lc - l0.astype(f32)
                for L in np.unique(l0[mini & win]):
                    sel = mini & (l0 == L)
                    L1 = min(int(L) + 1, nlev - 1)
                    s0_ = _bilinear(levels[int(L)], u, v)
                    s1_ = _bilinear(levels[L1], u, v)
                    luma = np.where(sel, s0_ + (s1_ - s0_) * fr, luma)
            colour = (np.clip(luma, 0, 1) * f32(255.0) + f32(0.5)).astype(np.uint32)
        frag = (depth.astype(np.uint64) << np.uint64(8)) | colour.astype(np.uint64)
        sub[win] = frag[win]


def render_stack(xyz, uv, levels, mvps, width, height):
    return np.stack([render_mesh(xyz, uv, levels, m, width, height) for m in np.asarray(mvps, f32).reshape(-1, 16)])
