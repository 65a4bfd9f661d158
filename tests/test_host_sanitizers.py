"""AddressSanitizer + UBSan run of the host-side C++ (YAML reader, grid arithmetic, strategy, map-file loaders) on hostile inputs.
CPU only (GPU sanitizers are not available on the pool)."""
import os
import subprocess

from conftest import ROOT


def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    srcs = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"),
            os.path.join(ROOT, "orbslam2_nmi_amd", "host", "nmi_driver.cpp"),
            os.path.join(ROOT, "orbslam2_nmi_amd", "host", "nmi_yaml.cpp"),
            os.path.join(ROOT, "orbslam2_nmi_amd", "host", "nmi_map.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orbslam2_nmi_amd", "host"),
                           *srcs, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TMPDIR=str(tmp_path)))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host sanitize ok" in r.stdout


def test_search_kernel_cpp_class(tmp_path):
    """The C++ class with the reference's interface, compiled with plain g++ (also under ASan/UBSan) and linked to the library."""
    from orbslam2_nmi_amd import build as nmi_build
    exe = tmp_path / "search_kernel_class"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orbslam2_nmi_amd", "host"),
                           os.path.join(ROOT, "tests", "native", "search_kernel_class.cpp"),
                           os.path.join(ROOT, "orbslam2_nmi_amd", "host", "nmi_driver.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "search kernel class ok" in r.stdout
