"""CPU test: host/nmi_rating.hpp -- NmiObjects::rating as a float****** view over the flat table nmi_search_grid writes, and
helperFunctions::find_max_elements with the reference's signature (localization.hpp:36,
localization.cpp:185-210, helperFunctions.cpp:50-103) -- compiled with plain g++ under ASan / UBSan against the C host library
and checked against a literal restatement of the reference's two passes (tests/native/rating_table.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rating_view_and_find_max_elements_with_the_references_signature(tmp_path):
    exe = tmp_path / "rating_table"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orbslam2_nmi_amd", "host"),
                           os.path.join(ROOT, "tests", "native", "rating_table.cpp"),
                           os.path.join(ROOT, "orbslam2_nmi_amd", "host", "nmi_driver.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rating table ok" in r.stdout
