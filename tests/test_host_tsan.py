"""The host side of the drop-in under ThreadSanitizer (ref: src/System.cc:365-375 starts Tracking, LocalMapping
and LoopClosing, each with matchers of its own over the same key frames).  tests/native/test_threads_tsan = host/*.cc + the
three-thread schedule of tests/native/test_threads_dropin.cpp (eight resident sets per thread so that every round evicts,
DropResidentSets() mid-run and every 97th round) + tests/native/mock_orbhip.cc in place of liborbhip.so, built with
-fsanitize=thread.  No GPU: the mock answers every search with a deterministic function of the data it was handed, aborts when a
context is entered by two threads at once, and keeps the resident sets' table with api_sets.hip's rules -- so a result that
differs from the single-threaded one is a host-side mix-up, and a data race in the host library is ThreadSanitizer's to report."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def test_three_matcher_threads_under_thread_sanitizer():
    subprocess.check_call(["make", "-C", NATIVE, "test_threads_tsan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 second_deadlock_stack=1")
    p = subprocess.run([os.path.join(NATIVE, "test_threads_tsan"), "250"], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
    assert p.returncode == 0 and "all ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    first = p.stdout.splitlines()[0]
    calls, evictions = int(first.split(";")[1].split()[0]), int(first.split(",")[-1].split()[0])
    assert calls > 2000 and evictions >= 250, first          # the schedule reaches the library and evicts every round
    for name in ("T SearchByBoW(KF,F)", "T SearchByProjection(F,F)", "M SearchForTriangulation", "M Fuse", "L SearchByBoW(KF,KF)",
                 "L SearchBySim3"):
        assert name + ": ok" in p.stdout, p.stdout
