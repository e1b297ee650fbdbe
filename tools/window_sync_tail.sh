#!/bin/bash
# round 5: what the closing synchronize of a K-step window waits for (deliveries vs chains in flight)
OUT=${1:-gpurun_out/r5b}; mkdir -p $OUT
for K in 20 192; do
  SPP_BENCH_TAIL=1 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > $OUT/tail_k$K.json 2> $OUT/tail_k$K.err
  grep "\[bench\] window" $OUT/tail_k$K.err
  python - $OUT/tail_k$K.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["windows"]["ms_per_step_all"], d["roofline"]["avg_launch_ms"])
PY
done
