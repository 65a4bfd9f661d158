"""ctypes view of the C ABI in include/nmi_hip.h (libnmi_hip.so).

There is no CPU fallback here: if the HIP library is missing or cannot be loaded this module raises,
and every entry point requires device (torch CUDA/HIP) tensors.  torch is used only to own device
memory and streams; the compute is the hand-written HIP in csrc/.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MODE_ENMI = 0  # Thirdparty/CUDA_Functions/kernel.cuh:22
MODE_SUC = 1   # Thirdparty/CUDA_Functions/kernel.cuh:23

NMI_OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_UNSUPPORTED = -2
ERR_NO_DEVICE = -3
ERR_NOT_READY = -4

# Every symbol include/nmi_hip.h declares; tests check that the library exports all of them.
EXPORTED_SYMBOLS = (
    "nmi_params_default", "nmi_create", "nmi_destroy", "nmi_set_stream", "nmi_synchronize", "nmi_eval_pair", "nmi_eval_pairs", "nmi_eval_pair_debug",
    "nmi_search_grid", "nmi_search_grid_shard", "nmi_search_grid_block", "nmi_warp_homographies", "nmi_warp_stack", "nmi_render_mvp", "nmi_render_points", "nmi_level_create", "nmi_level_create_mesh", "nmi_level_run", "nmi_level_copy_outputs", "nmi_level_destroy", "nmi_texture_create",
    "nmi_texture_destroy", "nmi_render_mesh", "nmi_stream_create", "nmi_stream_destroy",
    "nmi_stream_submit", "nmi_stream_wait", "nmi_stream_keep_ratings", "nmi_stream_copy_ratings", "nmi_key_pack", "nmi_key_unpack", "nmi_search_grid_rccl", "nmi_search_grid_block_rccl",
    "nmi_rccl_unique_id", "nmi_rccl_comm_init", "nmi_rccl_comm_destroy", "nmi_set_profiling", "nmi_last_kernel_ms",
    "nmi_set_option", "nmi_copy_term_table", "nmi_abi_version", "nmi_error_string", "nmi_last_error_detail", "nmi_get_info", "nmi_last_content", "nmi_sort_points", "nmi_sort_triangles",
    "nmi_split_status", "nmi_pix_status", "nmi_level_create_block", "nmi_level_create_mesh_block", "nmi_level_run_rccl", "nmi_stream_submit_block",
)


class NmiParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("bins", C.c_int32), ("mode", C.c_int32),
        ("use_bg", C.c_int32), ("render_bottom_up", C.c_int32), ("device", C.c_int32),
        ("max_candidates", C.c_int32), ("stream", C.c_void_p), ("reserved", C.c_int32 * 8),
    ]


class RenderParams(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("point_size", C.c_float)]


class NmiError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"{what} failed: {code} ({error_string(code)}) {detail}".strip())


_lib = None


def library_path():
    return _build.LIB


def load_library(build_if_missing=False):
    """Loads libnmi_hip.so.  Raises if it is not there (the product has no other compute path)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if build_if_missing:
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -m orbslam2_nmi_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the NMI path.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i32, i64p, f32p, u64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_uint64)
    lib.nmi_params_default.argtypes = [C.POINTER(NmiParams), i32, i32]
    lib.nmi_create.argtypes = [C.POINTER(NmiParams), C.POINTER(vp)]
    lib.nmi_destroy.argtypes = [vp]
    lib.nmi_set_stream.argtypes = [vp, vp]
    lib.nmi_synchronize.argtypes = [vp]
    lib.nmi_eval_pair.argtypes = [vp, vp, vp, f32p]
    lib.nmi_eval_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i32, f32p]
    lib.nmi_eval_pair_debug.argtypes = [vp, vp, vp, f32p, vp, vp, vp, vp]
    lib.nmi_search_grid.argtypes = [vp, vp, i32, vp, i32, vp, i64p, f32p]
    lib.nmi_search_grid_shard.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, u64p]
    lib.nmi_search_grid_block.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, vp, u64p]
    lib.nmi_warp_homographies.argtypes = [C.POINTER(C.c_double), C.POINTER(i32), f32p, C.POINTER(C.c_double)]
    lib.nmi_warp_stack.argtypes = [vp, vp, C.POINTER(C.c_double), i32, vp]
    lib.nmi_render_mvp.argtypes = [C.POINTER(RenderParams), f32p, f32p, f32p, f32p, f32p]
    lib.nmi_render_points.argtypes = [vp, vp, vp, C.c_int64, f32p, i32, C.c_float, vp]
    lib.nmi_texture_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.nmi_texture_destroy.argtypes = [vp]
    lib.nmi_render_mesh.argtypes = [vp, vp, vp, C.c_int64, vp, f32p, i32, vp]
    lib.nmi_level_create.argtypes = [vp, vp, vp, C.c_int64, vp, i32, i32, C.c_float, C.POINTER(vp)]
    lib.nmi_level_create_mesh.argtypes = [vp, vp, vp, C.c_int64, vp, vp, i32, i32, C.POINTER(vp)]
    lib.nmi_level_run.argtypes = [vp, f32p, C.POINTER(C.c_double), i64p, f32p]
    lib.nmi_level_create_block.argtypes = [vp, vp, vp, C.c_int64, vp, i32, i32, i32, i32, i32, i32, C.c_float, C.POINTER(vp)]
    lib.nmi_level_create_mesh_block.argtypes = [vp, vp, vp, C.c_int64, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.nmi_level_run_rccl.argtypes = [vp, f32p, C.POINTER(C.c_double), vp, i64p, f32p]
    lib.nmi_stream_submit_block.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(C.c_double), i32, i32, i32, vp, i64p]
    lib.nmi_level_copy_outputs.argtypes = [vp, vp, vp, vp]
    lib.nmi_level_destroy.argtypes = [vp]
    lib.nmi_stream_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.nmi_stream_destroy.argtypes = [vp]
    lib.nmi_stream_submit.argtypes = [vp, vp, i32, vp, C.POINTER(C.c_double), i32, i64p]
    lib.nmi_stream_wait.argtypes = [vp, C.c_int64, i64p, f32p]
    lib.nmi_stream_keep_ratings.argtypes = [vp, i32]
    lib.nmi_stream_copy_ratings.argtypes = [vp, C.c_int64, f32p, C.c_int64]
    lib.nmi_key_pack.argtypes = [C.c_float, C.c_int64]
    lib.nmi_key_pack.restype = C.c_uint64
    lib.nmi_key_unpack.argtypes = [C.c_uint64, i64p, f32p]
    lib.nmi_search_grid_rccl.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, i64p, f32p]
    lib.nmi_search_grid_block_rccl.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, vp, i64p, f32p]
    lib.nmi_rccl_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    lib.nmi_rccl_comm_init.argtypes = [vp, C.POINTER(C.c_uint8), i32, i32, C.POINTER(vp)]
    lib.nmi_rccl_comm_destroy.argtypes = [vp]
    lib.nmi_set_option.argtypes = [vp, i32, C.c_int64]
    lib.nmi_set_profiling.argtypes = [vp, i32]
    lib.nmi_last_kernel_ms.argtypes = [vp, f32p]
    lib.nmi_copy_term_table.argtypes = [vp, f32p, C.c_int64]
    lib.nmi_error_string.argtypes = [C.c_int]
    lib.nmi_error_string.restype = C.c_char_p
    lib.nmi_last_error_detail.argtypes = [vp]
    lib.nmi_last_error_detail.restype = C.c_char_p
    lib.nmi_get_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.nmi_last_content.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.nmi_split_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.nmi_pix_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.nmi_sort_points.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    lib.nmi_sort_triangles.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    _lib = lib
    return lib


def error_string(code):
    return load_library().nmi_error_string(int(code)).decode()


def key_pack(score, index):
    return int(load_library().nmi_key_pack(float(score), int(index)))


def key_unpack(key):
    idx, sc = C.c_int64(0), C.c_float(0)
    load_library().nmi_key_unpack(C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.byref(idx), C.byref(sc))
    return int(idx.value), np.float32(sc.value)


def warp_homographies(K, num_warp_xyz, step_rad_xyz):
    """K*Rz*Ry*Rx*K^-1 per warp cell (image.cpp:76-107) -> float64 [Wn, 3, 3], w = (wz*ny + wy)*nx + wx."""
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    num = (C.c_int32 * 3)(*[int(n) for n in num_warp_xyz])
    step = (C.c_float * 3)(*[float(s) for s in step_rad_xyz])
    wn = int(np.prod([int(n) for n in num_warp_xyz]))
    out = np.zeros((wn, 3, 3), np.float64)
    rc = load_library().nmi_warp_homographies(K.ctypes.data_as(C.POINTER(C.c_double)), num, step,
                                              out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != NMI_OK:
        raise NmiError(rc, "nmi_warp_homographies")
    return out


def render_mvp(rp, cam_pos, cam_look_at, cam_up, translation):
    """Projection * glm::lookAt(pos + t, look_at + t, up) as rendering.hpp:196-202,547-553 -> float32 [16], column-major."""
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    out = (C.c_float * 16)()
    rc = load_library().nmi_render_mvp(C.byref(rp), f3(cam_pos), f3(cam_look_at), f3(cam_up), f3(translation), out)
    if rc != NMI_OK:
        raise NmiError(rc, "nmi_render_mvp")
    return np.array(out, np.float32)


def _dev_u8(t, ndim, what):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{what} must be a device (HIP) torch tensor; the NMI path has no CPU implementation")
    if t.dtype != torch.uint8 or t.dim() != ndim or not t.is_contiguous():
        raise TypeError(f"{what} must be a contiguous uint8 tensor with {ndim} dims, got {t.dtype} {tuple(t.shape)}")
    return t


class NmiContext:
    """nmi_ctx wrapper.  Mirrors the per-search objects of the reference (NmiObjects' buffers +
    the CUDA scratch of kernel.cu:59-73) as one persistent workspace."""

    # {option: value} applied to every new context (tests use it to run whole suites with one kernel selection)
    default_options = {}

    def __init__(self, width, height, bins=256, mode=MODE_SUC, use_bg=True, render_bottom_up=True, device=None,
                 stream=None, max_candidates=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: orbslam2_nmi_amd needs an AMD GPU (gfx950)")
        self._lib = load_library()
        self.width, self.height = int(width), int(height)
        self.bins, self.mode, self.use_bg, self.render_bottom_up = bins, mode, bool(use_bg), bool(render_bottom_up)
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        p = NmiParams()
        self._check(self._lib.nmi_params_default(C.byref(p), self.width, self.height), "nmi_params_default")
        p.bins, p.mode, p.use_bg, p.render_bottom_up = int(bins), int(mode), int(bool(use_bg)), int(bool(render_bottom_up))
        p.device = self.device.index
        p.max_candidates = int(max_candidates)
        p.stream = stream
        self._h = C.c_void_p()
        rc = self._lib.nmi_create(C.byref(p), C.byref(self._h))
        if rc != NMI_OK:
            self._h = C.c_void_p()
            raise NmiError(rc, "nmi_create")
        for opt, val in type(self).default_options.items():
            self.set_option(opt, val)

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != NMI_OK:
            detail = self._lib.nmi_last_error_detail(self._h).decode() if getattr(self, "_h", None) else ""
            raise NmiError(rc, what, detail)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.nmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle):
        """Run on this hipStream_t (e.g. torch.cuda.Stream().cuda_stream); None = the context's own stream.

        The legacy default stream has handle 0, which the C ABI reads as "own stream": callers that need stream
        ordering with torch work (collectives, tensor reads) must make a non-default torch stream current and pass it."""
        if stream_handle == 0:
            raise ValueError("handle 0 is the legacy default stream; use a torch.cuda.Stream() (non-default) instead")
        self._check(self._lib.nmi_set_stream(self._h, C.c_void_p(stream_handle)), "nmi_set_stream")
        self._bound_stream = stream_handle

    def _order_after_torch(self):
        """Inputs produced by torch ops must be complete before this context's stream reads them: nothing to do when
        the context runs on torch's current stream, otherwise wait for that stream."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_bound_stream", None) != cur.cuda_stream:
            cur.synchronize()

    OPT_HIST_VARIANT, OPT_PHASE_MASK, OPT_WORKGROUPS, OPT_RESULT_PATH, OPT_XCD_TILING, OPT_TILE_QUEUE = 1, 2, 3, 4, 5, 6
    OPT_SPLIT, OPT_WAIT_MODE, OPT_STAMPS, OPT_SPLIT_PIXELS, OPT_CLIP_QUEUE = 7, 8, 9, 10, 11
    OPT_PIX_OWNER_BIAS = 14
    OPT_CONTENT_PATH, OPT_FEWLEVELS_BINS = 12, 13

    def set_option(self, option, value):
        self._check(self._lib.nmi_set_option(self._h, int(option), int(value)), "nmi_set_option")

    def synchronize(self):
        self._check(self._lib.nmi_synchronize(self._h), "nmi_synchronize")

    def set_profiling(self, on):
        self._check(self._lib.nmi_set_profiling(self._h, int(bool(on))), "nmi_set_profiling")

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.nmi_last_kernel_ms(self._h, C.byref(ms)), "nmi_last_kernel_ms")
        return float(ms.value)

    def info(self):
        cu, wg, lds = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.nmi_get_info(self._h, C.byref(cu), C.byref(wg), C.byref(lds)), "nmi_get_info")
        return {"compute_units": cu.value, "workgroups_per_launch": wg.value, "lds_bytes": lds.value}

    def last_content(self):
        """How the most recent search was scored -> {"few_levels": bool, "nr": int, "nw": int} (see nmi_last_content)."""
        f, r, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.nmi_last_content(self._h, C.byref(f), C.byref(r), C.byref(w)), "nmi_last_content")
        return {"few_levels": bool(f.value), "nr": r.value, "nw": w.value}

    def split_status(self):
        """Split-kernel liveness -> {"timeouts", "cooldown_calls_left", "next_cooldown", "last_launch_parts"} (nmi_split_status)."""
        v = [C.c_int32(0) for _ in range(4)]
        self._check(self._lib.nmi_split_status(self._h, *[C.byref(x) for x in v]), "nmi_split_status")
        return dict(zip(("timeouts", "cooldown_calls_left", "next_cooldown", "last_launch_parts"), (x.value for x in v)))

    def pix_status(self):
        """Pixel-range kernel for mid-size grids -> {"last_launch_ranges", "healed"} (nmi_pix_status; waits for the stream)."""
        v = [C.c_int32(0) for _ in range(2)]
        self._check(self._lib.nmi_pix_status(self._h, *[C.byref(x) for x in v]), "nmi_pix_status")
        return dict(zip(("last_launch_ranges", "healed"), (x.value for x in v)))

    def term_table(self):
        """The per-count entropy-term table (NMI.cu:242-263 evaluated once per possible count) as numpy float32 [W*H+1]."""
        out = np.zeros(self.width * self.height + 1, np.float32)
        self._check(self._lib.nmi_copy_term_table(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                    "nmi_copy_term_table")
        return out

    def _img(self, t, what):
        t = _dev_u8(t, 2, what)
        if tuple(t.shape) != (self.height, self.width):
            raise ValueError(f"{what} is {tuple(t.shape)}, context is {(self.height, self.width)}")
        return t

    def _stack(self, t, what):
        t = _dev_u8(t, 3, what)
        if tuple(t.shape[1:]) != (self.height, self.width):
            raise ValueError(f"{what} is {tuple(t.shape)}, context images are {(self.height, self.width)}")
        return t

    # -- the path -------------------------------------------------------------------------------
    def eval_pair(self, render, warped):
        """CUDAF::NMIWithCuda_noMask (kernel.cu:49-114) for one (render, warped frame) pair -> float32 score."""
        r, w = self._img(render, "render"), self._img(warped, "warped")
        out = C.c_float(0)
        self._order_after_torch()
        self._check(self._lib.nmi_eval_pair(self._h, r.data_ptr(), w.data_ptr(), C.byref(out)), "nmi_eval_pair")
        return np.float32(out.value)

    def eval_pairs(self, renders, warps):
        """n independent (render, warped) pairs (lists of device images) in as few launches as possible -> float32 [n]."""
        assert len(renders) == len(warps)
        n = len(renders)
        rp = (C.c_void_p * max(n, 1))(*[self._img(r, "render").data_ptr() for r in renders])
        wp = (C.c_void_p * max(n, 1))(*[self._img(w, "warped").data_ptr() for w in warps])
        out = np.zeros(n, np.float32)
        self._order_after_torch()
        self._check(self._lib.nmi_eval_pairs(self._h, rp, wp, n, out.ctypes.data_as(C.POINTER(C.c_float))), "nmi_eval_pairs")
        return out

    def eval_pair_debug(self, render, warped):
        """-> (score, joint[256,256] u32 (render x warped), hist_render[256], hist_warped[256], sums[3]) as numpy."""
        import torch
        r, w = self._img(render, "render"), self._img(warped, "warped")
        joint = torch.zeros(65536, dtype=torch.int32, device=self.device)
        h1 = torch.zeros(256, dtype=torch.int32, device=self.device)
        h2 = torch.zeros(256, dtype=torch.int32, device=self.device)
        sums = torch.zeros(3, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        out = C.c_float(0)
        self._check(self._lib.nmi_eval_pair_debug(self._h, r.data_ptr(), w.data_ptr(), C.byref(out), joint.data_ptr(),
                                                  h1.data_ptr(), h2.data_ptr(), sums.data_ptr()), "nmi_eval_pair_debug")
        u32 = lambda t: t.cpu().numpy().view(np.uint32)
        return np.float32(out.value), u32(joint).reshape(256, 256), u32(h1), u32(h2), sums.cpu().numpy()

    def sort_points(self, xyz, red):
        """Point cloud (device float32 [N,3], [N]) -> copies in Morton order (nmi_sort_points): same renders, faster."""
        import torch
        assert xyz.is_cuda and red.is_cuda and xyz.dtype == torch.float32 and red.dtype == torch.float32
        xyz, red = xyz.contiguous(), red.contiguous()
        n = xyz.shape[0]
        assert tuple(xyz.shape) == (n, 3) and red.numel() == n
        xo, ro = torch.empty_like(xyz), torch.empty_like(red)
        self._order_after_torch()
        self._check(self._lib.nmi_sort_points(self._h, xyz.data_ptr(), red.data_ptr(), n, xo.data_ptr(), ro.data_ptr()), "nmi_sort_points")
        return xo, ro

    def sort_triangles(self, xyz, uv):
        """Triangle soup (device float32 [3T,3] corners, [3T,2] texture coordinates) -> copies in Morton order of the centroids."""
        import torch
        assert xyz.is_cuda and uv.is_cuda and xyz.dtype == torch.float32 and uv.dtype == torch.float32
        xyz, uv = xyz.contiguous(), uv.contiguous()
        t = xyz.shape[0] // 3
        assert tuple(xyz.shape) == (3 * t, 3) and tuple(uv.shape) == (3 * t, 2)
        xo, uo = torch.empty_like(xyz), torch.empty_like(uv)
        self._order_after_torch()
        self._check(self._lib.nmi_sort_triangles(self._h, xyz.data_ptr(), uv.data_ptr(), t, xo.data_ptr(), uo.data_ptr()), "nmi_sort_triangles")
        return xo, uo

    def warp_stack(self, frame, homographies, out=None, sync=True):
        """Image::calculateWarping (image.cpp:115-128) on the device: frame [H,W] u8 + forward homographies [Wn,3,3]
        (float64, host) -> warp stack [Wn,H,W] u8 (device).  Enqueued on the context's stream."""
        import torch
        f = self._img(frame, "frame")
        m = np.ascontiguousarray(homographies, np.float64).reshape(-1, 9)
        wn = m.shape[0]
        if out is None:
            out = torch.empty((wn, self.height, self.width), dtype=torch.uint8, device=self.device)
        o = self._stack(out, "out")
        if o.shape[0] != wn:
            raise ValueError("out has the wrong number of warps")
        self._order_after_torch()  # `out` / `frame` may come from torch's stream
        self._check(self._lib.nmi_warp_stack(self._h, f.data_ptr(), m.ctypes.data_as(C.POINTER(C.c_double)), wn,
                                             o.data_ptr()), "nmi_warp_stack")
        if sync:
            self.synchronize()
        return out

    def render_points(self, xyz, red, mvps, point_size, out=None, sync=True):
        """Rendering<4>::renderToTextureOnGPU without OpenGL: device float32 xyz [N,3] + red [N], host MVPs [S,16]
        (render_mvp) -> render stack [S,H,W] u8 on the device, bottom-up rows, background 255."""
        import torch
        for t, shape in ((xyz, 2), (red, 1)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != shape:
                raise TypeError("xyz must be a contiguous float32 device tensor [N,3], red one of [N]")
        m = np.ascontiguousarray(mvps, np.float32).reshape(-1, 16)
        S = m.shape[0]
        if out is None:
            out = torch.empty((S, self.height, self.width), dtype=torch.uint8, device=self.device)
        o = self._stack(out, "out")
        self._order_after_torch()
        self._check(self._lib.nmi_render_points(self._h, xyz.data_ptr(), red.data_ptr(), xyz.shape[0],
                                                m.ctypes.data_as(C.POINTER(C.c_float)), S, float(point_size), o.data_ptr()),
                    "nmi_render_points")
        if sync:
            self.synchronize()
        return out

    def render_mesh(self, xyz, uv, texture, mvps, out=None, sync=True):
        """Rendering<1>::renderToTextureOnGPU without OpenGL: device float32 corner arrays xyz [3T,3], uv [3T,2], an
        NmiTexture, host MVPs [S,16] -> render stack [S,H,W] u8 on the device (bottom-up rows, background 255)."""
        import torch
        for t, cols in ((xyz, 3), (uv, 2)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != cols:
                raise TypeError("xyz / uv must be contiguous float32 device tensors [3T,3] / [3T,2]")
        if xyz.shape[0] != uv.shape[0] or xyz.shape[0] % 3:
            raise ValueError("xyz and uv must hold three corners per triangle")
        m = np.ascontiguousarray(mvps, np.float32).reshape(-1, 16)
        S = m.shape[0]
        if out is None:
            out = torch.empty((S, self.height, self.width), dtype=torch.uint8, device=self.device)
        o = self._stack(out, "out")
        self._order_after_torch()
        self._check(self._lib.nmi_render_mesh(self._h, xyz.data_ptr(), uv.data_ptr(), xyz.shape[0] // 3, texture._h,
                                              m.ctypes.data_as(C.POINTER(C.c_float)), S, o.data_ptr()), "nmi_render_mesh")
        if sync:
            self.synchronize()
        return out

    def search_grid(self, render_stack, warp_stack, ratings=None):
        """Candidate loop + arg-max (Tracking.cc:1879-1905,1952).  -> (best linear index w*S+s, best score).

        ratings: optional device float32 tensor [Wn, S] that receives the full rating table."""
        rs, ws = self._stack(render_stack, "render_stack"), self._stack(warp_stack, "warp_stack")
        S, Wn = rs.shape[0], ws.shape[0]
        rp = self._ratings_ptr(ratings, Wn, S)
        idx, sc = C.c_int64(0), C.c_float(0)
        self._order_after_torch()  # the stacks (and a pre-filled ratings tensor) may come from torch's stream
        self._check(self._lib.nmi_search_grid(self._h, rs.data_ptr(), S, ws.data_ptr(), Wn, rp, C.byref(idx), C.byref(sc)),
                    "nmi_search_grid")
        return int(idx.value), np.float32(sc.value)

    def bind_search(self, render_stack, warp_stack, ratings=None):
        """The arguments of search_grid converted ONCE -> a callable () -> (best linear index, best score) that makes the blocking
        nmi_search_grid call and nothing else.  For loops that repeat one search many times (bench.py's timed region): the checks
        and conversions of search_grid cost the interpreter a few microseconds per call, which is harness, not library.  The
        stacks must already be complete on the device (the caller synchronises once before the loop) and stay alive."""
        rs, ws = self._stack(render_stack, "render_stack"), self._stack(warp_stack, "warp_stack")
        S, Wn = rs.shape[0], ws.shape[0]
        rp = self._ratings_ptr(ratings, Wn, S)
        idx, sc = C.c_int64(0), C.c_float(0)
        pidx, psc, fn, handle = C.byref(idx), C.byref(sc), self._lib.nmi_search_grid, self._h
        prs, pws, cS, cWn = C.c_void_p(rs.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_int32(S), C.c_int32(Wn)
        prp = C.c_void_p(rp) if rp is not None else None
        check = self._check
        self._order_after_torch()

        def call(_keep=(rs, ws, ratings)):
            rc = fn(handle, prs, cS, pws, cWn, prp, pidx, psc)
            if rc != NMI_OK:
                check(rc, "nmi_search_grid")
            return idx.value, sc.value
        return call

    def search_grid_shard(self, render_stack, s_offset, s_total, warp_stack, ratings=None, key_out=None, blocking=True,
                          w_offset=0, wn_total=None):
        """One rank's part of a sharded search.  -> packed key (int) if blocking else None.

        key_out: optional device int64/uint64 tensor of one element that receives the key (for a collective).
        w_offset / wn_total: position of the given warps in the global warp axis when that axis is sharded too."""
        rs, ws = self._stack(render_stack, "render_stack"), self._stack(warp_stack, "warp_stack")
        S, Wn = rs.shape[0], ws.shape[0]
        rp = self._ratings_ptr(ratings, Wn, S)
        kp = None
        if key_out is not None:
            if not key_out.is_cuda or key_out.numel() != 1 or key_out.element_size() != 8:
                raise TypeError("key_out must be a one-element 64-bit device tensor")
            kp = key_out.data_ptr()
        hk = C.c_uint64(0)
        self._order_after_torch()
        self._check(self._lib.nmi_search_grid_block(self._h, rs.data_ptr(), S, int(s_offset), int(s_total), ws.data_ptr(),
                                                    Wn, int(w_offset), int(Wn + w_offset if wn_total is None else wn_total), rp, kp,
                                                    C.byref(hk) if blocking else None),
                    "nmi_search_grid_block")
        return int(hk.value) if blocking else None

    def _ratings_ptr(self, ratings, Wn, S):
        import torch
        if ratings is None:
            return None
        if (not ratings.is_cuda or ratings.dtype != torch.float32 or not ratings.is_contiguous()
                or ratings.numel() != Wn * S):
            raise TypeError(f"ratings must be a contiguous float32 device tensor with {Wn * S} elements")
        return ratings.data_ptr()

    # -- RCCL -----------------------------------------------------------------------------------
    def rccl_comm_init(self, unique_id: bytes, rank, nranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        comm = C.c_void_p()
        self._check(self._lib.nmi_rccl_comm_init(self._h, buf, rank, nranks, C.byref(comm)), "nmi_rccl_comm_init")
        return comm

    def search_grid_rccl(self, render_stack, s_offset, s_total, warp_stack, comm, ratings=None, w_offset=0, wn_total=None):
        """One rank's block of a sharded search + the RCCL MAX all-reduce of the packed winner -> global (index, score)."""
        rs, ws = self._stack(render_stack, "render_stack"), self._stack(warp_stack, "warp_stack")
        S, Wn = rs.shape[0], ws.shape[0]
        idx, sc = C.c_int64(0), C.c_float(0)
        self._order_after_torch()
        self._check(self._lib.nmi_search_grid_block_rccl(self._h, rs.data_ptr(), S, int(s_offset), int(s_total), ws.data_ptr(),
                                                         Wn, int(w_offset), int(Wn + w_offset if wn_total is None else wn_total),
                                                         self._ratings_ptr(ratings, Wn, S), comm, C.byref(idx), C.byref(sc)),
                    "nmi_search_grid_block_rccl")
        return int(idx.value), np.float32(sc.value)


class NmiTexture:
    """nmi_texture wrapper: RGB8 image as handed to glTexImage2D -> mip chain -> per-level luma on the device."""

    def __init__(self, ctx, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self.ctx, self._lib = ctx, ctx._lib
        self._h = C.c_void_p()
        ctx._check(self._lib.nmi_texture_create(ctx._h, rgb.ctypes.data_as(C.c_void_p), rgb.shape[1], rgb.shape[0],
                                                C.byref(self._h)), "nmi_texture_create")

    def close(self):
        if self._h and self._h.value:
            self._lib.nmi_texture_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class NmiLevel:
    """nmi_level wrapper: cloud + frame -> renders, warps, search, winner as one captured HIP graph."""

    def __init__(self, ctx, xyz, red, frame, S, Wn, point_size, texture=None, block=None):
        """Point cloud: xyz [N,3], red [N], point_size.  Textured mesh: pass texture=NmiTexture, xyz [3T,3] corner
        positions and `red` = uv [3T,2] (point_size is ignored).
        block = (s_offset, S_total, w_offset, Wn_total): this level is one rank's block (S views x Wn warps, either may be 0)
        of a level sharded over ranks (nmi_level_create_block); winners then carry global indices."""
        self.ctx, self._lib = ctx, ctx._lib
        self._keep = (xyz, red, frame, texture)  # the graph holds their device addresses
        self.S, self.Wn = int(S), int(Wn)
        self._h = C.c_void_p()
        ctx._order_after_torch()
        if block is None:
            block = (0, self.S, 0, self.Wn)
        so, st, wo, wt = (int(v) for v in block)
        if texture is None:
            ctx._check(self._lib.nmi_level_create_block(ctx._h, xyz.data_ptr(), red.data_ptr(), xyz.shape[0], frame.data_ptr(), self.S, so,
                                                        st, self.Wn, wo, wt, float(point_size), C.byref(self._h)), "nmi_level_create_block")
        else:
            ctx._check(self._lib.nmi_level_create_mesh_block(ctx._h, xyz.data_ptr(), red.data_ptr(), xyz.shape[0] // 3, texture._h,
                                                             frame.data_ptr(), self.S, so, st, self.Wn, wo, wt, C.byref(self._h)),
                       "nmi_level_create_mesh_block")

    def _params(self, mvps, homographies):
        m = np.ascontiguousarray(mvps, np.float32).reshape(-1)
        h = np.ascontiguousarray(homographies, np.float64).reshape(-1)
        assert m.size == self.S * 16 and h.size == self.Wn * 9
        return m, h, (m.ctypes.data_as(C.POINTER(C.c_float)) if m.size else None), (h.ctypes.data_as(C.POINTER(C.c_double)) if h.size else None)

    def run(self, mvps, homographies):
        """mvps [S,16], homographies [Wn,3,3] of THIS block -> (global index, score) of the block's winner."""
        m, h, mp, hp = self._params(mvps, homographies)
        idx, sc = C.c_int64(0), C.c_float(0)
        self.ctx._check(self._lib.nmi_level_run(self._h, mp, hp, C.byref(idx), C.byref(sc)), "nmi_level_run")
        return int(idx.value), np.float32(sc.value)

    def bind(self, mvps, homographies, comm=None):
        """Parameters converted ONCE -> a callable that replays the level with them: () -> (global index, score).  For loops that
        replay a few fixed parameter sets many times (bench.py --config e2e): the conversions of run() cost the interpreter ~20 us a
        call, a tenth of the level."""
        m, h, mp, hp = self._params(mvps, homographies)
        idx, sc = C.c_int64(0), C.c_float(0)
        pidx, psc, lib, handle, check = C.byref(idx), C.byref(sc), self._lib, self._h, self.ctx._check

        def replay(_keep=(m, h)):   # (the arrays behind the pointers stay alive with the closure)
            if comm is None:
                check(lib.nmi_level_run(handle, mp, hp, pidx, psc), "nmi_level_run")
            else:
                check(lib.nmi_level_run_rccl(handle, mp, hp, comm, pidx, psc), "nmi_level_run_rccl")
            return idx.value, sc.value
        return replay

    def run_rccl(self, mvps, homographies, comm):
        """The same + the level's MAX all-reduce over the RCCL communicator -> the LEVEL's winner, on every rank."""
        m, h, mp, hp = self._params(mvps, homographies)
        idx, sc = C.c_int64(0), C.c_float(0)
        self.ctx._check(self._lib.nmi_level_run_rccl(self._h, mp, hp, comm, C.byref(idx), C.byref(sc)), "nmi_level_run_rccl")
        return int(idx.value), np.float32(sc.value)

    def outputs(self):
        """-> (renders [S,H,W] u8, warps [Wn,H,W] u8, ratings [Wn,S] f32) of the latest run, as numpy (host copies)."""
        h, w = self.ctx.height, self.ctx.width
        r = np.empty((self.S, h, w), np.uint8)
        v = np.empty((self.Wn, h, w), np.uint8)
        t = np.empty((self.Wn, self.S), np.float32)
        self.ctx._check(self._lib.nmi_level_copy_outputs(self._h, r.ctypes.data, v.ctypes.data, t.ctypes.data), "nmi_level_copy_outputs")
        return r, v, t

    def close(self):
        if self._h and self._h.value:
            self._lib.nmi_level_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class NmiStream:
    """nmi_stream wrapper: double-buffered keyframe pipeline (H2D of render stacks beside the search)."""

    def __init__(self, ctx, max_S, max_Wn, depth=2):
        self.ctx = ctx
        self._lib = ctx._lib
        self._h = C.c_void_p()
        ctx._check(self._lib.nmi_stream_create(ctx._h, int(max_S), int(max_Wn), int(depth), C.byref(self._h)),
                   "nmi_stream_create")
        self._keep = {}

    def submit(self, render_stack_host, frame_host=None, homographies=None, block=None, comm=None):
        """render_stack_host / frame_host: pinned CPU uint8 torch tensors; homographies: float64 [Wn,3,3] (with a frame).
        block = (s_offset, S_total, w_offset, Wn_total): the stack and the homographies are this rank's block of a level
        sharded over ranks (nmi_stream_submit_block); comm: RCCL communicator whose ranks all-reduce the level's key."""
        rs = render_stack_host
        if rs.is_cuda or rs.dtype.__str__() != "torch.uint8" or not rs.is_contiguous():
            raise TypeError("render_stack_host must be a contiguous CPU uint8 tensor (pinned for overlap)")
        fp, mp, wn, m = None, None, 0, None
        if frame_host is not None:
            m = np.ascontiguousarray(homographies, np.float64).reshape(-1, 9)
            fp, mp, wn = frame_host.data_ptr(), m.ctypes.data_as(C.POINTER(C.c_double)), m.shape[0]
        t = C.c_int64(-1)
        if block is None and comm is None:
            self.ctx._check(self._lib.nmi_stream_submit(self._h, rs.data_ptr(), rs.shape[0], fp, mp, wn, C.byref(t)),
                            "nmi_stream_submit")
        else:
            if block is None:
                # a communicator without a block: a blockless level cannot say how many warps a frame-less submission reuses
                raise ValueError("NmiStream.submit(comm=...) needs block=(s_offset, S_total, w_offset, Wn_total): the position of this "
                                 "rank's views and warps in the level the ranks all-reduce over")
            so, st, wo, wt = (int(v) for v in block)
            self.ctx._check(self._lib.nmi_stream_submit_block(self._h, rs.data_ptr() if rs.shape[0] else None, rs.shape[0], so, st, fp, mp,
                                                              wn, wo, wt, comm, C.byref(t)), "nmi_stream_submit_block")
        self._keep[t.value] = (rs, frame_host, m)  # keep host buffers alive until the ticket completes
        return int(t.value)

    def keep_ratings(self, on=True):
        self.ctx._check(self._lib.nmi_stream_keep_ratings(self._h, int(bool(on))), "nmi_stream_keep_ratings")

    def ratings(self, ticket, Wn, S):
        """Rating table [Wn, S] of a ticket that has been waited for (needs keep_ratings())."""
        t = np.empty((Wn, S), np.float32)
        self.ctx._check(self._lib.nmi_stream_copy_ratings(self._h, int(ticket), t.ctypes.data_as(C.POINTER(C.c_float)), t.size),
                        "nmi_stream_copy_ratings")
        return t

    def wait(self, ticket):
        idx, sc = C.c_int64(0), C.c_float(0)
        self.ctx._check(self._lib.nmi_stream_wait(self._h, int(ticket), C.byref(idx), C.byref(sc)), "nmi_stream_wait")
        self._keep.pop(ticket, None)
        return int(idx.value), np.float32(sc.value)

    def close(self):
        if self._h and self._h.value:
            self._lib.nmi_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def rccl_unique_id():
    buf = (C.c_uint8 * 128)()
    rc = load_library().nmi_rccl_unique_id(buf)
    if rc != NMI_OK:
        raise NmiError(rc, "nmi_rccl_unique_id")
    return bytes(buf)


def rccl_comm_destroy(comm):
    load_library().nmi_rccl_comm_destroy(comm)
