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
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DEMO OK" in r.stdout and "NmiKernel:" in r.stdout
    # part C of the demo: the reference's unchanged per-candidate call site (src/Tracking.cc:1886-1894) through the shim,
    # one blocking call per candidate at 640x480.  BASELINE.json's target for the path is 50,000 evals/s; measured
    # 50.6-53.1 k box to box in round 2 and 52.5-52.8 k in round 3 (profiles/r03_b/shim_rate.txt; bench.py reports the figure
    # of the box it runs on as "call_site").  The floor asserted here leaves room for a slow box and a busy host core.
    rate = float(re.search(r"SHIM_EVALS_PER_S (\d+)", r.stdout).group(1))
    batched = float(re.search(r"SHIM_BATCHED_EVALS_PER_S (\d+)", r.stdout).group(1))
    assert rate >= 46000, rate
    assert batched >= 3 * rate, (rate, batched)


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
