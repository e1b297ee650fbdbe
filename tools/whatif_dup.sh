#!/bin/bash
# round 5: in-situ cost of each idempotent chain kernel (SPP_WHATIF_DUP launches it twice; the step's increase is its cost)
OUT=${1:-gpurun_out/r5c}; mkdir -p $OUT
K=${K:-192}
for rep in 1 2; do
for dup in none count pick flag rows "count,pick,flag,rows"; do
  tag=$(echo $dup | tr ',' '_')
  SPP_WHATIF_DUP=$dup python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > $OUT/dup_${tag}_$rep.json 2> $OUT/dup_${tag}_$rep.err
  python - $OUT/dup_${tag}_$rep.json $dup <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:.4f}  deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f}  windows {d['windows']['ms_per_step_all']}")
PY
done
done
