"""The oracle's OpenCV-side kernels against the independent implementations available on this image
(tools/crosscheck_opencv.py).  scikit-image lives under /opt/conda/bin/python3.9 on the
build box and on the GPU box; the test is skipped where that interpreter is missing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "crosscheck_opencv.py")
CONDA = "/opt/conda/bin/python3.9"


def _run(py):
    return subprocess.run([py, TOOL], capture_output=True, text=True, timeout=600)


def test_numpy_rows_with_the_test_interpreter(oracle):
    r = _run(sys.executable)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fastAtan2" in r.stdout and "MISMATCH" not in r.stdout
    # round 3: remap / undistortPoints / initUndistortRectifyMap against scipy (the test interpreter has it)
    out = r.stdout
    assert out.count("remap INTER_LINEAR, BORDER_CONSTANT (orbo_remap_linear_u8)") == 2 and "mode=grid-constant" in out
    assert "five fixed-point iterations" in out and "within 150 px of the principal point" in out
    assert "initUndistortRectifyMap (orbo_init_undistort_rectify_map)" in out
    rows = [l for l in out.splitlines() if l.startswith("| remap") or l.startswith("| undistortPoints") or l.startswith("| initUndistort")]
    assert len(rows) == 6 and all(l.rstrip().endswith("| ok |") for l in rows)


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no /opt/conda/bin/python3.9 (scikit-image) on this box")
def test_skimage_rows(oracle):
    probe = subprocess.run([CONDA, "-c", "import skimage, numpy"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("scikit-image not importable under " + CONDA)
    r = _run(CONDA)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "MISMATCH" not in out
    # the exact rows really ran
    assert out.count("FAST-9/16 corner set") == 6 and "0 differing (pixel, t) pairs" in out
    assert "rBRIEF pattern table T0 (256 pairs) | scikit-image" in out and "| equal | ok |" in out
    # round 5: the steering arithmetic of rBRIEF (src/ORBextractor.cc:110-149) against scikit-image's own rotated-pattern loop on the
    # oracle's keypoints, angles and blurred levels: every differing bit (if any) is a rounding tie of a rotated coordinate
    steer = [l for l in out.splitlines() if l.startswith("| rBRIEF steering")]
    assert len(steer) == 1 and " 0 non-tie differences" in steer[0] and steer[0].rstrip().endswith("| ok |")
