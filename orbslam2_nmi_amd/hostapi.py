"""ctypes view of include/nmi_host.h: the host-side search driver (grid descriptor, arg-max rule, pose update,
coarse-to-fine strategy) that mirrors the reference's NmiSearchKernel / helperFunctions / Tracking NMI functions.
No GPU is needed for anything in here."""
import ctypes as C

import numpy as np

from .capi import load_library

AXES = ("synthX", "synthY", "synthZ", "warpX", "warpY", "warpZ")
MAX_ITER = 16

EXPORTED_SYMBOLS = (
    "nmi_properties_default", "nmi_sk_init", "nmi_sk_reset", "nmi_sk_is_middle", "nmi_sk_resize", "nmi_sk_candidates",
    "nmi_sk_format", "nmi_sk_linear_index", "nmi_sk_set_best_from_index", "nmi_find_max_elements",
    "nmi_calculate_translation", "nmi_calculate_relocalization", "nmi_mat4_inverse", "nmi_relocalize_with_strategy",
    "nmi_config_parse", "nmi_config_load", "nmi_map_load_obj", "nmi_map_load_xyz", "nmi_map_load_bmp", "nmi_map_free",
)


class SearchKernel(C.Structure):
    _fields_ = [("num", C.c_int32 * 6), ("step", C.c_float * 6), ("best", C.c_int32 * 6), ("nmi", C.c_float)]

    @classmethod
    def make(cls, num, step, best=None, nmi=0.0):
        k = cls()
        _lib().nmi_sk_init(C.byref(k))
        k.num[:] = list(num)
        k.step[:] = list(step)
        if best is not None:
            k.best[:] = list(best)
        k.nmi = nmi
        return k

    def as_dict(self):
        return {"num": list(self.num), "step": [np.float32(s) for s in self.step], "best": list(self.best),
                "nmi": np.float32(self.nmi)}


class Properties(C.Structure):
    _fields_ = [("max_iteration_count", C.c_int32), ("reloc_frequency", C.c_int32), ("step_factor", C.c_float),
                ("min_kernel_rotation", C.c_double), ("min_kernel_translation", C.c_double), ("use_bg", C.c_int32)]


class StrategyInput(C.Structure):
    _fields_ = [("Tcw", C.c_float * 16), ("distance_since_last", C.c_float * 3), ("rotation_since_last", C.c_float * 3),
                ("not_initialized", C.c_int32), ("nmi_threshold", C.c_float), ("initial", SearchKernel)]


class StrategyOutput(C.Structure):
    _fields_ = [("Tcw", C.c_float * 16), ("relocalized", C.c_int32), ("failed", C.c_int32), ("iterations", C.c_int32),
                ("stop_reason", C.c_int32), ("reverted_to_previous", C.c_int32), ("nmi_threshold_used", C.c_float),
                ("kernel", SearchKernel), ("last_kernel", SearchKernel), ("per_iteration", SearchKernel * MAX_ITER)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("initial", SearchKernel), ("nmi_threshold", C.c_float), ("init_offset", C.c_int32),
                ("has_init1", C.c_int32), ("has_init2", C.c_int32), ("init1", C.c_float * 16), ("init2", C.c_float * 16),
                ("render_point_size", C.c_float), ("render_near", C.c_float), ("render_far", C.c_float),
                ("render_object", C.c_char * 512), ("render_texture", C.c_char * 512), ("render_cloud", C.c_char * 512),
                ("render_offset", C.c_char * 512)]

    def K(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])


EVAL_GRID_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SearchKernel), C.POINTER(C.c_float), C.POINTER(C.c_int64),
                           C.POINTER(C.c_float))

_configured = False


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        skp, f32p = C.POINTER(SearchKernel), C.POINTER(C.c_float)
        lib.nmi_properties_default.argtypes = [C.POINTER(Properties)]
        lib.nmi_properties_default.restype = None
        for name in ("nmi_sk_init", "nmi_sk_reset"):
            getattr(lib, name).argtypes = [skp]
            getattr(lib, name).restype = None
        lib.nmi_sk_is_middle.argtypes = [skp]
        lib.nmi_sk_resize.argtypes = [skp, C.POINTER(Properties)]
        lib.nmi_sk_resize.restype = None
        lib.nmi_sk_candidates.argtypes = [skp]
        lib.nmi_sk_candidates.restype = C.c_int64
        lib.nmi_sk_format.argtypes = [skp, C.c_char_p, C.c_size_t]
        lib.nmi_sk_linear_index.argtypes = [skp, C.POINTER(C.c_int32)]
        lib.nmi_sk_linear_index.restype = C.c_int64
        lib.nmi_sk_set_best_from_index.argtypes = [skp, C.c_int64, C.c_float]
        lib.nmi_find_max_elements.argtypes = [f32p, C.c_int64, C.POINTER(C.c_int64), C.c_int64, f32p]
        lib.nmi_find_max_elements.restype = C.c_int64
        lib.nmi_calculate_translation.argtypes = [f32p, skp, C.c_int32, C.c_int32, C.c_int32, f32p]
        lib.nmi_calculate_relocalization.argtypes = [f32p, skp, f32p]
        lib.nmi_mat4_inverse.argtypes = [f32p, f32p]
        lib.nmi_relocalize_with_strategy.argtypes = [C.POINTER(StrategyInput), C.POINTER(Properties), EVAL_GRID_FN,
                                                     C.c_void_p, C.POINTER(StrategyOutput)]
        lib.nmi_config_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Config)]
        lib.nmi_config_load.argtypes = [C.c_char_p, C.POINTER(Config)]
        fpp, i64p = C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64)
        lib.nmi_map_load_obj.argtypes = [C.c_char_p, fpp, fpp, i64p]
        lib.nmi_map_load_xyz.argtypes = [C.c_char_p, C.c_char_p, fpp, fpp, fpp, i64p]
        lib.nmi_map_load_bmp.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.nmi_map_free.argtypes = [C.c_void_p]
        lib.nmi_map_free.restype = None
        _configured = True
    return lib


def properties_default():
    p = Properties()
    _lib().nmi_properties_default(C.byref(p))
    return p


def _f32(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    assert a.size == n
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def is_middle(k):
    return bool(_lib().nmi_sk_is_middle(C.byref(k)))


def resize(k, props=None):
    _lib().nmi_sk_resize(C.byref(k), C.byref(props) if props is not None else None)
    return k


def candidates(k):
    return int(_lib().nmi_sk_candidates(C.byref(k)))


def fmt(k):
    buf = C.create_string_buffer(512)
    n = _lib().nmi_sk_format(C.byref(k), buf, 512)
    return buf.raw[:n].decode()


def linear_index(k, idx6):
    return int(_lib().nmi_sk_linear_index(C.byref(k), (C.c_int32 * 6)(*idx6)))


def set_best_from_index(k, index, score):
    return _lib().nmi_sk_set_best_from_index(C.byref(k), int(index), float(score))


def find_max_elements(ratings, cap=None):
    r = np.ascontiguousarray(ratings, np.float32).reshape(-1)
    cap = r.size if cap is None else cap
    ties = np.zeros(max(cap, 1), np.int64)
    mx = C.c_float(0)
    n = _lib().nmi_find_max_elements(r.ctypes.data_as(C.POINTER(C.c_float)), r.size,
                                     ties.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(mx))
    return ties[:min(n, cap)].tolist(), int(n), np.float32(mx.value)


def calculate_translation(Twc, k, sx, sy, sz):
    t, tp = _f32(Twc, 16)
    out = np.zeros(3, np.float32)
    assert _lib().nmi_calculate_translation(tp, C.byref(k), sx, sy, sz, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out


def calculate_relocalization(Twc, k):
    t, tp = _f32(Twc, 16)
    out = np.zeros(16, np.float32)
    assert _lib().nmi_calculate_relocalization(tp, C.byref(k), out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out.reshape(4, 4)


def mat4_inverse(m):
    a, ap = _f32(m, 16)
    out = np.zeros(16, np.float32)
    assert _lib().nmi_mat4_inverse(ap, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out.reshape(4, 4)


def config_parse(text):
    """Camera.* / NMI.* keys of a reference settings file (cv::FileStorage YAML subset) -> Config; raises on errors."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    cfg = Config()
    rc = _lib().nmi_config_parse(raw, len(raw), C.byref(cfg))
    if rc != 0:
        raise ValueError(f"nmi_config_parse failed: {rc}")
    return cfg


def config_load(path):
    cfg = Config()
    rc = _lib().nmi_config_load(str(path).encode(), C.byref(cfg))
    if rc != 0:
        raise ValueError(f"nmi_config_load({path}) failed: {rc}")
    return cfg


def _take(ptr, shape, dtype):
    """Copy of a malloc'ed C array as numpy, and the C array released."""
    n = int(np.prod(shape))
    out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).reshape(shape) if n else np.zeros(shape, dtype)
    _lib().nmi_map_free(C.cast(ptr, C.c_void_p))
    return out


def load_obj(path):
    """loadOBJ (objloader.cpp:140-224) -> (xyz float32 [3T,3], uv float32 [3T,2]): the per-corner arrays of nmi_render_mesh."""
    xyz, uv, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int64()
    rc = _lib().nmi_map_load_obj(str(path).encode(), C.byref(xyz), C.byref(uv), C.byref(n))
    if rc != 0:
        raise ValueError(f"nmi_map_load_obj({path}) failed: {rc}")
    return _take(xyz, (n.value, 3), np.float32), _take(uv, (n.value, 2), np.float32)


def load_xyz(path, offset_path):
    """loadXYZ (objloader.cpp:226-264) -> (xyz float32 [N,3], red float32 [N], rgb float32 [N,3]); red is what nmi_render_points takes."""
    xyz, red, rgb, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int64()
    rc = _lib().nmi_map_load_xyz(str(path).encode(), str(offset_path).encode(), C.byref(xyz), C.byref(red), C.byref(rgb), C.byref(n))
    if rc != 0:
        raise ValueError(f"nmi_map_load_xyz({path}) failed: {rc}")
    return _take(xyz, (n.value, 3), np.float32), _take(red, (n.value,), np.float32), _take(rgb, (n.value, 3), np.float32)


def load_bmp(path):
    """loadBMP_custom (texture.cpp:31-86) -> uint8 [H,W,3] in file order: the image nmi_texture_create takes."""
    rgb, w, h = C.POINTER(C.c_uint8)(), C.c_int32(), C.c_int32()
    rc = _lib().nmi_map_load_bmp(str(path).encode(), C.byref(rgb), C.byref(w), C.byref(h))
    if rc != 0:
        raise ValueError(f"nmi_map_load_bmp({path}) failed: {rc}")
    return _take(rgb, (h.value, w.value, 3), np.uint8)


def relocalize_with_strategy(Tcw, initial, eval_grid, distance=(0, 0, 0), rotation=(0, 0, 0), not_initialized=False,
                             nmi_threshold=0.1, props=None):
    """eval_grid(kernel: SearchKernel, Twc: np.ndarray[4,4]) -> (best_index, best_score).  Returns StrategyOutput."""
    inp = StrategyInput()
    inp.Tcw[:] = np.asarray(Tcw, np.float32).reshape(-1).tolist()
    inp.distance_since_last[:] = [float(x) for x in distance]
    inp.rotation_since_last[:] = [float(x) for x in rotation]
    inp.not_initialized = int(bool(not_initialized))
    inp.nmi_threshold = float(nmi_threshold)
    inp.initial = initial
    err = []

    def cb(_user, kp, twc, best_index, best_score):
        try:
            idx, sc = eval_grid(kp.contents, np.ctypeslib.as_array(twc, shape=(16,)).reshape(4, 4).copy())
            best_index[0] = int(idx)
            best_score[0] = float(sc)
            return 0
        except Exception as e:  # surfaced after the C call returns
            err.append(e)
            return -100

    out = StrategyOutput()
    rc = _lib().nmi_relocalize_with_strategy(C.byref(inp), C.byref(props) if props is not None else None,
                                             EVAL_GRID_FN(cb), None, C.byref(out))
    if err:
        raise err[0]
    if rc != 0:
        raise RuntimeError(f"nmi_relocalize_with_strategy failed: {rc}")
    return out
