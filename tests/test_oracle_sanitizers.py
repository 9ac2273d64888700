"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are
not available on this pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_runs_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c99", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "native", "oracle_asan_main.c"),
                           os.path.join(ROOT, "oracle", "orb_oracle.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle sanitizer run ok" in out.stdout
