#!/bin/bash
# round 5: scatter folded into the pick kernel (SPP_FUSE_SCATTER=0 never / 1 small hops / 2 every hop):
# parity suites, then lone chain (two streams and one) and the pipeline at K = 20 / 192, interleaved
set -e
OUT=${1:-gpurun_out/r5i}; mkdir -p $OUT
for mode in 1 2; do
  SPP_FUSE_SCATTER=$mode python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_random_graphs.py -x -q > $OUT/tests_fuse$mode.txt 2>&1 || { tail -40 $OUT/tests_fuse$mode.txt; exit 1; }
  tail -2 $OUT/tests_fuse$mode.txt
done
set +e
for rep in 1 2; do
for mode in 0 1 2; do
  echo "== SPP_FUSE_SCATTER=$mode rep $rep"
  SPP_FUSE_SCATTER=$mode WL=S-papers CHAIN_CFG=64,16 python tools/microbench.py chain 2>&1 | grep "chain only"
  SPP_FUSE_SCATTER=$mode SPP_WORK_STREAMS=1 WL=S-papers CHAIN_CFG=64,16 python tools/microbench.py chain 2>&1 | grep "chain only" | sed 's/^/  one stream: /'
  for K in 20 192; do
    SPP_FUSE_SCATTER=$mode python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > $OUT/bench_fuse${mode}_k${K}_$rep.json 2> $OUT/bench_fuse${mode}_k${K}_$rep.err
    python - $OUT/bench_fuse${mode}_k${K}_$rep.json $K <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"  K={sys.argv[2]:4s} ms/step {d['ms_per_step']:.4f} deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f} windows {d['windows']['ms_per_step_all']}")
PY
  done
done
done
