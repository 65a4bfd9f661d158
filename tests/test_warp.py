"""Warp-stack producer: host homographies vs image.cpp:76-107 semantics, the fp32 oracle vs a float64
bilinear model (CPU), and the HIP kernel vs the fp32 oracle (GPU, bit-exact).  Parity with OpenCV itself is unpinned
(oracle/warp_oracle_np.py header)."""
import numpy as np
import pytest

from orbslam2_nmi_amd import capi, synthetic as sy
from orbslam2_nmi_amd import build as nmi_build
from oracle import warp_oracle_np as wo


@pytest.fixture(scope="module", autouse=True)
def _built():
    nmi_build.build()


@pytest.mark.parametrize("counts", [(3, 3, 3), (1, 1, 1), (4, 2, 5), (2, 1, 3)])
def test_homographies_follow_image_cpp(counts):
    K = sy.intrinsics(640, 480)
    steps = (0.02, 0.03, 0.05)
    got = capi.warp_homographies(K, counts, steps)
    exp = sy.warp_homographies(K, counts, [float(np.float32(s)) for s in steps])  # independent python model
    assert got.shape == exp.shape and np.allclose(got, exp, rtol=0, atol=1e-9 * np.abs(exp).max())
    # odd counts: the centre cell is the identity; even counts: -(n-1)/2 truncates (image.cpp:77), so cell (n-1)//2 is
    nx, ny, nz = counts
    c = (((nz - 1) // 2) * ny + (ny - 1) // 2) * nx + (nx - 1) // 2
    assert np.allclose(got[c], np.eye(3), atol=1e-12)


def test_fp32_oracle_close_to_float64_model():
    B = sy.scene(320, 240, 2)
    Ms = sy.warp_homographies(sy.intrinsics(320, 240), (3, 3, 3), (0.02, 0.02, 0.05))
    for M in Ms[::5]:
        a = wo.warp_perspective(B, M).astype(int)
        b = sy.warp_perspective(B, M).astype(int)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 0.02
    assert (wo.warp_perspective(B, np.eye(3)) == B).all()                     # identity is exact
    assert np.allclose(wo.inverse_coeffs(Ms[3]), wo.inverse_coeffs_adjugate(Ms[3]), rtol=2e-7, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(640, 480), (333, 97), (64, 48)])
def test_gpu_warp_stack_bit_exact_vs_fp32_oracle(shape):
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = shape
    F = sy.camera_frame(sy.scene(w, h, 4), 5)
    Ms = capi.warp_homographies(sy.intrinsics(w, h), (3, 3, 3), (0.02, 0.02, 0.05))
    # plus a strong warp that pushes most of the frame out of view, and a singular-ish scale
    extra = np.array([[[1.3, 0.2, -40], [-0.1, 0.9, 25], [1e-4, -2e-4, 1.0]], [[0.5, 0, w / 2], [0, 0.5, h / 2], [0, 0, 1.0]]])
    Ms = np.concatenate([Ms, extra])
    with nmi.NmiContext(w, h) as ctx:
        got = ctx.warp_stack(torch.from_numpy(F).cuda(), Ms).cpu().numpy()
    exp = wo.warp_stack(F, Ms)
    assert got.shape == exp.shape
    assert (got == exp).all(), f"{(got != exp).sum()} pixels differ, max {np.abs(got.astype(int) - exp.astype(int)).max()}"
    c = 13
    assert (got[c] == F).all()  # identity warp reproduces the frame


@pytest.mark.gpu
def test_gpu_frame_to_winner_end_to_end():
    """frame -> device warp stack -> grid search: the planted centre cell wins, table equals the oracle's on the same stacks."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    from oracle import binding as oc
    w, h = 320, 240
    B = sy.scene(w, h, 1234)
    F = sy.camera_frame(B, 1235)
    rs = sy.render_stack(B, (3, 3, 3), bottom_up=True)
    Ms = capi.warp_homographies(sy.intrinsics(w, h), (3, 3, 3), (0.02, 0.02, 0.05))
    with nmi.NmiContext(w, h) as ctx:
        ws = ctx.warp_stack(torch.from_numpy(F).cuda(), Ms, sync=False)   # same stream as the search: no sync needed
        t = torch.zeros(27, 27, device="cuda")
        idx, best = ctx.search_grid(torch.from_numpy(rs).cuda(), ws, t)
    ro, io, bo = oc.search_grid(rs, ws.cpu().numpy(), threads=8)
    assert idx == io == 13 * 27 + 13 and abs(float(best) - float(bo)) <= 1e-5
    assert np.abs(t.cpu().numpy() - ro).max() <= 1e-5
