#!/bin/bash
OUT=gpurun_out/r04_frame3; mkdir -p $OUT
python -m pytest tests/test_frame_build.py tests/test_vocabulary.py tests/test_bench_shapes.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -8 > $OUT/tests.txt
bash tools/latency_native.sh 3000 > $OUT/latency_native.json 2> $OUT/latency_native.err
python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --verify 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'], d['verified_frames'])" > $OUT/bench.txt
cat $OUT/tests.txt $OUT/latency_native.json $OUT/bench.txt; tail -3 $OUT/latency_native.err
