"""numpy (fp32) restatement of the warp-stack producer -- TEST INFRASTRUCTURE ONLY.

What it restates: Image::calculateWarping, Thirdparty/Localization/image.cpp:115-128, i.e.
cv::cuda::warpPerspective(src, dst, M, size) with the defaults INTER_LINEAR / BORDER_CONSTANT(0) / forward matrix.
OpenCV 3.4.0 (CUDA_Functions.vcxproj / Localization.vcxproj dependency, not vendored in the reference, absent in this
image) does the arithmetic; this file follows its published device path from the library's sources as remembered:
invert M on the host in double, pass 9 floats, per destination pixel
    coeff = 1 / (c6*x + c7*y + c8);  xs = coeff * (c0*x + c1*y + c2),  ys = coeff * (c3*x + c4*y + c5)     (fp32)
    LinearFilter: x1 = floor(xs), y1 = floor(ys); out = s(y1,x1)*((x2-xs)*(y2-ys)) + s(y1,x2)*((xs-x1)*(y2-ys))
                  + s(y2,x1)*((x2-xs)*(ys-y1)) + s(y2,x2)*((xs-x1)*(ys-y1));  border taps = 0
    saturate_cast<uchar>: round to nearest even, clamp to [0, 255].
"PARITY UNPINNED": no OpenCV here and no fixture in the reference to check this against.
"""
import numpy as np

f32 = np.float32


def inverse_coeffs(M):
    """Host side of warpPerspective: invert the forward matrix in double, hand 9 floats to the device."""
    return np.linalg.inv(np.asarray(M, np.float64).reshape(3, 3)).astype(f32).reshape(9)


def inverse_coeffs_adjugate(M):
    """Same inverse written as adjugate / determinant in double -- the form the product's host code uses."""
    m = np.asarray(M, np.float64).reshape(9)
    det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6])
    inv = np.array([(m[4] * m[8] - m[5] * m[7]), (m[2] * m[7] - m[1] * m[8]), (m[1] * m[5] - m[2] * m[4]),
                    (m[5] * m[6] - m[3] * m[8]), (m[0] * m[8] - m[2] * m[6]), (m[2] * m[3] - m[0] * m[5]),
                    (m[3] * m[7] - m[4] * m[6]), (m[1] * m[6] - m[0] * m[7]), (m[0] * m[4] - m[1] * m[3])]) / det
    return inv.astype(f32)


def warp_perspective(img, M, coeffs=None):
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    c = inverse_coeffs_adjugate(M) if coeffs is None else np.asarray(coeffs, f32)
    yy, xx = np.mgrid[0:h, 0:w]
    fx, fy = xx.astype(f32), yy.astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        coeff = f32(1.0) / ((c[6] * fx + c[7] * fy) + c[8])
        xs = coeff * ((c[0] * fx + c[1] * fy) + c[2])
        ys = coeff * ((c[3] * fx + c[4] * fy) + c[5])
        inside = (xs > f32(-2)) & (xs < f32(w + 1)) & (ys > f32(-2)) & (ys < f32(h + 1))
        xs = np.where(inside, xs, f32(-10))
        ys = np.where(inside, ys, f32(-10))
        x1 = np.floor(xs).astype(np.int64)
        y1 = np.floor(ys).astype(np.int64)
        x2, y2 = x1 + 1, y1 + 1

        def tap(yi, xi):
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            return np.where(ok, img[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)], 0).astype(f32)

        ax2, ax1 = x2.astype(f32) - xs, xs - x1.astype(f32)
        ay2, ay1 = y2.astype(f32) - ys, ys - y1.astype(f32)
        acc = np.zeros((h, w), f32)
        acc = acc + tap(y1, x1) * (ax2 * ay2)
        acc = acc + tap(y1, x2) * (ax1 * ay2)
        acc = acc + tap(y2, x1) * (ax2 * ay1)
        acc = acc + tap(y2, x2) * (ax1 * ay1)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def warp_stack(frame, Ms):
    return np.stack([warp_perspective(frame, M) for M in Ms])
