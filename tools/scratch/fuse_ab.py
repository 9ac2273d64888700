import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_configs as bc
r = bc.run_fuse(640, 480, 1000, 256)
print("fuse keyframes/s %.0f" % r[0], r[1:])
