"""The C++ host programs on the drop-in boundary, no Python in the loop: examples/relocalize_demo.cpp (reference-style
classes + C ABI + shim, toy host renderer) and examples/level_pipeline.cpp (renders, warps and search on the device, one
HIP graph per strategy iteration).  CPU tier: they build and link.  GPU tier: each recovers a planted pose offset."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "examples", "relocalize_demo")
EXE_PIPELINE = os.path.join(ROOT, "examples", "level_pipeline")


def test_demo_builds():
    from orbslam2_nmi_amd import build as nmi_build
    nmi_build.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert os.access(EXE, os.X_OK) and os.access(EXE_PIPELINE, os.X_OK)


@pytest.mark.gpu
def test_demo_recovers_planted_offset():
    if not os.access(EXE, os.X_OK):  # normally prebuilt by __graft_entry__.build(); hipcc exists on the GPU box too
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # Part C of the demo: the reference's unchanged per-candidate call site (src/Tracking.cc:1886-1894) through the shim, one
    # blocking call per candidate at 640x480, then the two lines behind it (rating[..][..][..][..][..][..] = nmi; find_max_elements).
    # BASELINE.json's target for the path is 50,000 evals/s; this call site measures 48-53 k box to box (it straddles the target;
    # 200-230 k with BeginBatch / Flush around the warp loop) -- bench.py reports the figure of the box it runs on as "call_site"
    # with "target_met".  This is a CORRECTNESS test: the rates are printed, not asserted.
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DEMO OK" in r.stdout and "NmiKernel:" in r.stdout
    assert "find_max_elements: 1 winner(s)" in r.stdout and "tables equal" in r.stdout
    assert float(re.search(r"SHIM_EVALS_PER_S (\d+)", r.stdout).group(1)) > 0
    assert float(re.search(r"SHIM_BATCHED_EVALS_PER_S (\d+)", r.stdout).group(1)) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("map_args", [[], ["--mesh"], ["--mesh", "60x40"]], ids=["cloud", "mesh-120k", "mesh-4800"])
def test_device_pipeline_recovers_planted_offset(map_args):
    """Point cloud (nmi_prop_RENDER 4) and textured mesh (nmi_prop_RENDER 1, the reference's default) as the map."""
    if not os.access(EXE_PIPELINE, os.X_OK):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([EXE_PIPELINE, "20", *map_args], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PIPELINE OK" in r.stdout and "levels/s" in r.stdout


def test_pipeline_writes_map_files_the_loaders_read(tmp_path):
    """level_pipeline --write-files: its synthetic map as OBJ + BMP (or XYZ + offset) and a settings file in the reference's YAML
    format; the host library reads all of them back (no GPU involved)."""
    import numpy as np
    from orbslam2_nmi_amd import hostapi as HA
    if not os.access(EXE_PIPELINE, os.X_OK):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    mesh_dir, cloud_dir = tmp_path / "mesh", tmp_path / "cloud"
    mesh_dir.mkdir(), cloud_dir.mkdir()
    subprocess.check_call([EXE_PIPELINE, "--mesh", "12x8", "--write-files", str(mesh_dir)], stdout=subprocess.DEVNULL)
    subprocess.check_call([EXE_PIPELINE, "--density", "0.05", "--write-files", str(cloud_dir)], stdout=subprocess.DEVNULL)
    cfg = HA.config_load(mesh_dir / "settings.yaml")
    assert (cfg.width, cfg.height) == (848, 480) and list(cfg.initial.num) == [3] * 6 and cfg.render_object == b"map.obj"
    xyz, uv = HA.load_obj(mesh_dir / "map.obj")
    assert xyz.shape == (12 * 8 * 6, 3) and uv.shape == (12 * 8 * 6, 2)
    assert np.array_equal(uv[:3], np.array([[0, 0], [1 / 12, 1 / 8], [1 / 12, 0]], np.float32))   # corner order of the first triangle
    assert HA.load_bmp(mesh_dir / "map.bmp").shape == (1024, 2048, 3)
    cfg = HA.config_load(cloud_dir / "settings.yaml")
    assert cfg.render_cloud == b"map.xyz" and cfg.render_object == b""
    pts, red, _ = HA.load_xyz(cloud_dir / "map.xyz", cloud_dir / "map.offset")
    nu, nv = int(3 * 848 * np.float32(0.05)), int(3 * 480 * np.float32(0.05))
    assert pts.shape == (nu * nv, 3) and 5.0 < pts[:, 2].min() and pts[:, 2].max() < 15.0 and 0 <= red.min() and red.max() < 1


@pytest.mark.gpu
@pytest.mark.parametrize("map_args", [["--mesh", "60x40"], ["--density", "0.5"]], ids=["mesh-4800", "cloud"])
def test_device_pipeline_from_map_files(tmp_path, map_args):
    """The same relocalisation with camera, grid, render parameters and map taken from files: settings.yaml (nmi_config_load),
    OBJ + BMP or XYZ + offset (nmi_map_load_*).  The text round trip is exact, so the winner must be the in-memory run's."""
    if not os.access(EXE_PIPELINE, os.X_OK):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([EXE_PIPELINE, *map_args, "--write-files", str(tmp_path)], stdout=subprocess.DEVNULL)
    direct = subprocess.run([EXE_PIPELINE, "5", *map_args], capture_output=True, text=True, timeout=300)
    files = subprocess.run([EXE_PIPELINE, "5", "--files", str(tmp_path)], capture_output=True, text=True, timeout=300)
    print(files.stdout, files.stderr)
    assert files.returncode == 0 and "PIPELINE OK" in files.stdout, files.stdout + files.stderr
    assert direct.returncode == 0, direct.stdout + direct.stderr
    pick = lambda out: [l for l in out.splitlines() if l.startswith(("NmiKernel:", "relocalized="))]
    assert pick(files.stdout) == pick(direct.stdout) and len(pick(files.stdout)) >= 3
