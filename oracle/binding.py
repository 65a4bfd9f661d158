"""ctypes binding of oracle/libnmi_oracle.so -- TEST INFRASTRUCTURE ONLY (see nmi_oracle.c header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libnmi_oracle.so")
MODE_ENMI, MODE_SUC = 0, 1

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


def build(force=False):
    src = os.path.join(_HERE, "nmi_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnmi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        l.nmi_oracle_joint_hist.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u32p, _u32p, _u32p]
        l.nmi_oracle_joint_hist.restype = None
        l.nmi_oracle_bin_term.argtypes = [C.c_uint32, C.c_int]
        l.nmi_oracle_bin_term.restype = C.c_float
        l.nmi_oracle_score_from_hist.argtypes = [_u32p, _u32p, _u32p, C.c_int, C.c_int, _f32p]
        l.nmi_oracle_score_from_hist.restype = C.c_float
        l.nmi_oracle_eval_pair.argtypes = [_u8p, _u8p] + [C.c_int] * 6
        l.nmi_oracle_eval_pair.restype = C.c_float
        l.nmi_oracle_eval_pair_f64.argtypes = [_u8p, _u8p] + [C.c_int] * 6
        l.nmi_oracle_eval_pair_f64.restype = C.c_double
        l.nmi_oracle_find_max.argtypes = [_f32p, C.c_int64, _f32p]
        l.nmi_oracle_find_max.restype = C.c_int64
        l.nmi_oracle_search_grid.argtypes = [_u8p, C.c_int, _u8p, C.c_int] + [C.c_int] * 7 + [_f32p, _f32p]
        l.nmi_oracle_search_grid.restype = C.c_int64
        l.nmi_oracle_max_threads.restype = C.c_int
        l.nmi_oracle_set_term_mode.argtypes = [C.c_int]
        l.nmi_oracle_set_term_mode.restype = None
        l.nmi_oracle_get_term_mode.restype = C.c_int
        l.nmi_oracle_term_table.argtypes = [C.c_int, _f32p]
        l.nmi_oracle_term_table.restype = None
        _lib = l
    return _lib


TERM_LIBM, TERM_ROUNDED = 0, 1


class term_mode:
    """with term_mode(TERM_ROUNDED): ...  -- per-bin log2 evaluated as the correctly rounded fp32 value (see
    nmi_oracle_bin_term); the default TERM_LIBM is the host libm's log2f."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib().nmi_oracle_get_term_mode()
        lib().nmi_oracle_set_term_mode(self.mode)
        return self

    def __exit__(self, *exc):
        lib().nmi_oracle_set_term_mode(self.prev)


def rounded():
    return term_mode(TERM_ROUNDED)


def term_table(length):
    out = np.zeros(int(length) + 1, np.float32)
    lib().nmi_oracle_term_table(int(length), out.ctypes.data_as(_f32p))
    return out


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def joint_hist(render, warped, shift=0, use_bg=True, render_bottom_up=True):
    r, rp = _u8(render)
    w, wp = _u8(warped)
    assert r.shape == w.shape and r.ndim == 2
    j = np.zeros(65536, np.uint32)
    h1 = np.zeros(256, np.uint32)
    h2 = np.zeros(256, np.uint32)
    lib().nmi_oracle_joint_hist(rp, wp, r.shape[1], r.shape[0], shift, int(use_bg), int(render_bottom_up),
                                j.ctypes.data_as(_u32p), h1.ctypes.data_as(_u32p), h2.ctypes.data_as(_u32p))
    return j.reshape(256, 256), h1, h2


def score_from_hist(joint, h1, h2, length, mode=MODE_SUC):
    j = np.ascontiguousarray(joint, np.uint32).reshape(-1)
    h1 = np.ascontiguousarray(h1, np.uint32)
    h2 = np.ascontiguousarray(h2, np.uint32)
    sums = np.zeros(3, np.float32)
    s = lib().nmi_oracle_score_from_hist(j.ctypes.data_as(_u32p), h1.ctypes.data_as(_u32p), h2.ctypes.data_as(_u32p),
                                         int(length), mode, sums.ctypes.data_as(_f32p))
    return np.float32(s), sums


def eval_pair(render, warped, shift=0, use_bg=True, render_bottom_up=True, mode=MODE_SUC):
    r, rp = _u8(render)
    w, wp = _u8(warped)
    assert r.shape == w.shape and r.ndim == 2
    return np.float32(lib().nmi_oracle_eval_pair(rp, wp, r.shape[1], r.shape[0], shift, int(use_bg),
                                                 int(render_bottom_up), mode))


def eval_pair_f64(render, warped, shift=0, use_bg=True, render_bottom_up=True, mode=MODE_SUC):
    r, rp = _u8(render)
    w, wp = _u8(warped)
    return float(lib().nmi_oracle_eval_pair_f64(rp, wp, r.shape[1], r.shape[0], shift, int(use_bg),
                                                int(render_bottom_up), mode))


def find_max(ratings):
    r = np.ascontiguousarray(ratings, np.float32).reshape(-1)
    best = C.c_float(0)
    idx = lib().nmi_oracle_find_max(r.ctypes.data_as(_f32p), r.size, C.byref(best))
    return int(idx), np.float32(best.value)


def search_grid(render_stack, warp_stack, shift=0, use_bg=True, render_bottom_up=True, mode=MODE_SUC, threads=1):
    """-> (ratings[Wn, S] float32, best linear index w*S+s, best score)."""
    r, rp = _u8(render_stack)
    w, wp = _u8(warp_stack)
    assert r.ndim == 3 and w.ndim == 3 and r.shape[1:] == w.shape[1:]
    S, Wn = r.shape[0], w.shape[0]
    ratings = np.zeros((Wn, S), np.float32)
    best = C.c_float(0)
    idx = lib().nmi_oracle_search_grid(rp, S, wp, Wn, r.shape[2], r.shape[1], shift, int(use_bg),
                                       int(render_bottom_up), mode, threads, ratings.ctypes.data_as(_f32p),
                                       C.byref(best))
    return ratings, int(idx), np.float32(best.value)


def max_threads():
    return int(lib().nmi_oracle_max_threads())
