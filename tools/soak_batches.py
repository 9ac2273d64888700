#!/usr/bin/env python3
"""Random extraction configurations in batches of 2, 8 and 16 frames (the batch kernels and their schedule) against the oracle:
tests/test_rare_events.py's generator with a seed and a time budget of one's own.  usage: soak_batches.py seconds seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("vi-orb-slam-icra2018_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import test_rare_events as T
budget, seed = float(sys.argv[1]), int(sys.argv[2])
print("soak batches ok:", T.run_extraction_configs(10 ** 9, seed, budget), "(%d %d)" % (budget, seed))
