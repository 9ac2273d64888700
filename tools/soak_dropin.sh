# Soak of the LocalMapping / LoopClosing drop-ins (tests/native/test_fuse_dropin): N seeds x a few synthetic frames, every run
# self-checks the HIP path against the host restatement.  usage: bash tools/soak_dropin.sh [nseeds]
cd /root/repo
N=${1:-20}
mkdir -p /tmp/soakdrop
python - <<'PY'
import sys
sys.path.insert(0, "vi-orb-slam-icra2018_amd")
from orbhip import synth
for i, (w, h) in enumerate([(640, 480), (752, 480), (376, 241)]):
    open("/tmp/soakdrop/f%d_%dx%d.raw" % (i, w, h), "wb").write(synth.make_frames(300 + i, w, h, 1)[0].tobytes())
PY
ok=0; bad=0
for s in $(seq 1 $N); do
  for f in /tmp/soakdrop/*.raw; do
    wh=${f##*_}; wh=${wh%.raw}; w=${wh%x*}; h=${wh#*x}
    nf=$((700 + (s * 137) % 1400))
    if tests/native/test_fuse_dropin $w $h $nf $f $s > /tmp/soakdrop/out.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "FAILED seed $s $f nf $nf"; grep -v ": ok" /tmp/soakdrop/out.txt | head -5; fi
  done
done
echo "dropin soak: $ok runs ok, $bad failed"
