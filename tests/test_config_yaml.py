"""Config loader for the reference's YAML settings surface: Camera.* / NMI.* keys without OpenCV."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from orbslam2_nmi_amd import build as nmi_build
from orbslam2_nmi_amd import hostapi as H


@pytest.fixture(scope="module", autouse=True)
def _built():
    nmi_build.build()


def test_example_settings_file():
    cfg = H.config_load(os.path.join(GOLDEN, "settings_example.yaml"))
    assert (cfg.width, cfg.height) == (960, 540)
    assert (cfg.fx, cfg.fy, cfg.cx, cfg.cy) == (435.04593205, 435.04593205, 475.55781765, 274.7487729)
    assert list(cfg.initial.num) == [3, 3, 5, 3, 1, 3]           # includes the `key:value` (no blank) lines
    assert np.allclose(list(cfg.initial.step), [0.2, 0.2, 0.5, 0.02, 0.02, 0.05])
    assert list(cfg.initial.best) == [-1] * 6 and cfg.initial.nmi == 0   # NmiSearchKernel ctor state
    assert cfg.nmi_threshold == np.float32(0.1) and cfg.init_offset == 10
    assert cfg.has_init1 and cfg.has_init2
    assert np.allclose(np.array(cfg.init1).reshape(4, 4)[:3, 3], [-7.5, 2.25, 30.0])     # multi-line data block
    assert np.allclose(np.array(cfg.init2).reshape(4, 4)[0], [0.5, -0.5, 0.0, 1.0])
    assert (cfg.render_point_size, cfg.render_near, cfg.render_far) == (3.0, 5.0, 30.0)
    assert cfg.render_object.decode() == "D:/data/mesh/model # not a comment.obj"
    assert cfg.render_cloud.decode() == "cloud.xyz"
    assert np.allclose(cfg.K(), [[435.04593205, 0, 475.55781765], [0, 435.04593205, 274.7487729], [0, 0, 1]])


def test_errors():
    with pytest.raises(ValueError, match="-5"):
        H.config_load("/nonexistent/settings.yaml")
    with pytest.raises(ValueError, match="-3"):
        H.config_parse("%YAML:1.0\nNMI.SynthNumX: 3\n")
    base = open(os.path.join(GOLDEN, "settings_example.yaml")).read()
    with pytest.raises(ValueError, match="-4"):
        H.config_parse(base.replace("NMI.WarpStepZ: 0.05", ""))
    with pytest.raises(ValueError, match="-2"):
        H.config_parse(base.replace("cols: 4\n    dt: f\n    data: [1.0,", "cols: 5\n    dt: f\n    data: [1.0,", 1))
    cfg = H.config_parse(base.replace("NMI.Treshold: 0.1", "").replace("NMI.Offset: 10", ""))
    assert cfg.nmi_threshold == 0 and cfg.init_offset == 0   # optional keys default to 0 like an empty cv::FileNode


def test_reference_settings_files_if_present():
    """The reference's own example settings (only where the reference tree is mounted; never on the GPU box)."""
    files = sorted(glob.glob("/root/reference/Examples/Monocular/*.yaml"))
    if not files:
        pytest.skip("reference tree not mounted")
    for f in files:
        if "NMI.SynthNumX" not in open(f, encoding="latin-1").read():
            with pytest.raises(ValueError):   # a plain ORB-SLAM2 settings file without the NMI surface (ETH.yaml)
                H.config_load(f)
            continue
        cfg = H.config_load(f)
        assert cfg.width > 0 and cfg.height > 0 and cfg.fx > 0
        assert all(n >= 1 for n in cfg.initial.num) and all(s > 0 for s in cfg.initial.step)
        assert cfg.has_init1 and cfg.has_init2 and cfg.nmi_threshold > 0
    eth = H.config_load("/root/reference/Examples/Monocular/ETH_small.yaml")
    assert (eth.width, eth.height) == (960, 540) and list(eth.initial.num) == [3] * 6   # the 3^6 = 729 default grid
