"""numpy (fp32) restatement of the point-cloud render-stack producer -- TEST INFRASTRUCTURE ONLY.

Restates Rendering<4>::renderToTextureOnGPU (Thirdparty/Localization/rendering.hpp:530-630) + shaders/ShadingWithColor.*:
clear colour 1.0 (:533), gl_Position = MVP * vec4(p, 1), GL_POINTS of glPointSize (:307), depth test GL_LESS (:294-297),
red channel into a GL_RED 8-bit texture (:347).  The rasteriser belongs to the OpenGL driver, which the reference does
not contain; the point rules are those of the OpenGL 3.3 specification (section 3.4.1, non-antialiased points) and a
24-bit depth buffer.  "PARITY UNPINNED": nothing in the reference tree (and no GL here) to check these pixels against.
"""
import numpy as np

f32 = np.float32


def look_at(eye, center, up):
    """glm::lookAt (right-handed), float64 model used only to cross-check the product's fp32 matrix."""
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[0, 3], V[1, 3], V[2, 3] = -s @ eye, -u @ eye, f @ eye
    return V


def projection(fx, fy, cx, cy, zn, zf):
    """rendering.hpp:196-202 (glm columns written out as a conventional row-major 4x4)."""
    P = np.zeros((4, 4))
    P[0, 0] = fx / (-cx)
    P[1, 1] = fy / (-cy)
    P[2, 2] = (zn + zf) / (zn - zf)
    P[3, 2] = -1.0
    P[2, 3] = 2 * zn * zf / (zn - zf)
    return P


def render_points(xyz, red, mvp_colmajor, width, height, point_size):
    """One view.  mvp_colmajor: float32[16], glm layout m[c*4+r].  Returns uint8 [H, W], bottom-up rows."""
    m = np.asarray(mvp_colmajor, f32)
    x, y, z = (np.asarray(xyz, f32)[:, k] for k in range(3))
    cx = (m[0] * x + m[4] * y) + (m[8] * z + m[12])
    cy = (m[1] * x + m[5] * y) + (m[9] * z + m[13])
    cz = (m[2] * x + m[6] * y) + (m[10] * z + m[14])
    cw = (m[3] * x + m[7] * y) + (m[11] * z + m[15])
    keep = (cw > 0) & (cx >= -cw) & (cx <= cw) & (cy >= -cw) & (cy <= cw) & (cz >= -cw) & (cz <= cw)
    cx, cy, cz, cw = cx[keep], cy[keep], cz[keep], cw[keep]
    iw = f32(1.0) / cw                                  # the perspective divide as one reciprocal and three products (splat_point)
    xw = (cx * iw * f32(0.5) + f32(0.5)) * f32(width)
    yw = (cy * iw * f32(0.5) + f32(0.5)) * f32(height)
    zw = cz * iw * f32(0.5) + f32(0.5)
    depth = (zw * f32(16777215.0) + f32(0.5)).astype(np.uint32)
    colour = (np.clip(np.asarray(red, f32)[keep], 0, 1) * f32(255.0) + f32(0.5)).astype(np.uint32)
    frag = (depth << np.uint32(8)) | colour
    size = max(1, min(64, int(np.floor(f32(point_size) + f32(0.5)))))
    if size & 1:
        x0 = np.floor(xw).astype(np.int64) - (size - 1) // 2
        y0 = np.floor(yw).astype(np.int64) - (size - 1) // 2
    else:
        x0 = np.floor(xw + f32(0.5)).astype(np.int64) - size // 2
        y0 = np.floor(yw + f32(0.5)).astype(np.int64) - size // 2
    zbuf = np.full(width * height, 0xFFFFFFFF, np.uint32)
    for dy in range(size):
        for dx in range(size):
            px, py = x0 + dx, y0 + dy
            ok = (px >= 0) & (px < width) & (py >= 0) & (py < height)
            np.minimum.at(zbuf, (py[ok] * width + px[ok]), frag[ok])
    return (zbuf & np.uint32(0xFF)).astype(np.uint8).reshape(height, width)


def render_stack(xyz, red, mvps, width, height, point_size):
    return np.stack([render_points(xyz, red, m, width, height, point_size) for m in np.asarray(mvps, f32).reshape(-1, 16)])
