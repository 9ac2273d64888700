"""Public declarations of the drop-in classes against the reference's headers (build container only; tools/diff_dropin_headers.py
reads /root/reference at run time and stores nothing from it; skipped where the reference is absent, e.g. on the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_public_declaration_of_the_reference_classes_has_a_dropin():
    if not os.path.isdir("/root/reference/include"):
        pytest.skip("reference tree absent")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diff_dropin_headers.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    lines = [l for l in out.stdout.splitlines() if " public " in l]
    assert len(lines) == 2 and all("missing  0" in l for l in lines), out.stdout
