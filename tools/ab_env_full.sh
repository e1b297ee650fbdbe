#!/bin/bash
# usage: tools/ab_env_full.sh <outdir> <VAR> <value A> <value B> -- parity suites under B, then interleaved
# lone chain (two streams / one) and pipeline at K = 192 / 20 under A and B, two repetitions
OUT=$1; VAR=$2; A=$3; B=$4; mkdir -p $OUT
env $VAR=$B timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_random_graphs.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -1 $OUT/tests.txt
for rep in 1 2; do
for v in $A $B; do
  echo "== $VAR=$v rep $rep"
  env $VAR=$v WL=S-papers CHAIN_CFG=64,16 timeout -k 10 200 python tools/microbench.py chain 2>&1 | grep "chain only"
  env $VAR=$v SPP_WORK_STREAMS=1 WL=S-papers CHAIN_CFG=64,16 timeout -k 10 200 python tools/microbench.py chain 2>&1 | grep "chain only" | sed 's/^/  one stream: /'
  for K in 192 20; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > $OUT/bench_${v}_k${K}_$rep.json 2> $OUT/bench_${v}_k${K}_$rep.err || { tail -5 $OUT/bench_${v}_k${K}_$rep.err; exit 1; }
    python - $OUT/bench_${v}_k${K}_$rep.json $K <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"  K={sys.argv[2]:4s} ms/step {d['ms_per_step']:.4f} deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f} windows {d['windows']['ms_per_step_all']}")
PY
  done
done
done
