import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle"),
          os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import orb_oracle_py
    orb_oracle_py.build()
    return orb_oracle_py


# ---- counts of the rare-event tests (tests/test_rare_events.py), printed in the terminal summary so that the driver's log shows
# how many frame pairs / configurations the run compared with the oracle ----
_TALLY = {}


@pytest.fixture
def tally():
    def add(name, n):
        _TALLY[name] = _TALLY.get(name, 0) + int(n)
    return add


def pytest_terminal_summary(terminalreporter):
    if _TALLY:
        terminalreporter.write_line("orbhip rare-event coverage: " + "; ".join("%s = %d" % kv for kv in sorted(_TALLY.items())))
