import sys, os
sys.path.insert(0, 'vi-orb-slam-icra2018_amd')
import numpy as np, torch
from orbhip import synth, extractor
fr = synth.make_frames(1000, 640, 480, 8)
ex = extractor.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=8)
out = ex.extract_batch(fr)
print("done", len(out))
