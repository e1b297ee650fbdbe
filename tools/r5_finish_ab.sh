#!/bin/bash
# round 5: k_hop_finish (flag + rows as one pass, decoupled look-back) -- parity suites, then lone chain and pipeline A/B
OUT=${1:-gpurun_out/r5j}; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_random_graphs.py -x -q > $OUT/tests_finish.txt 2>&1
rc=$?; tail -15 $OUT/tests_finish.txt
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
for fin in 0 1; do
  echo "== SPP_FINISH=$fin rep $rep"
  SPP_FINISH=$fin WL=S-papers CHAIN_CFG=64,16 timeout -k 10 200 python tools/microbench.py chain 2>&1 | grep "chain only"
  SPP_FINISH=$fin SPP_WORK_STREAMS=1 WL=S-papers CHAIN_CFG=64,16 timeout -k 10 200 python tools/microbench.py chain 2>&1 | grep "chain only" | sed 's/^/  one stream: /'
  for K in 20 192; do
    SPP_FINISH=$fin timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > $OUT/bench_fin${fin}_k${K}_$rep.json 2> $OUT/bench_fin${fin}_k${K}_$rep.err || { tail -5 $OUT/bench_fin${fin}_k${K}_$rep.err; exit 1; }
    python - $OUT/bench_fin${fin}_k${K}_$rep.json $K <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"  K={sys.argv[2]:4s} ms/step {d['ms_per_step']:.4f} deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f} windows {d['windows']['ms_per_step_all']}")
PY
  done
done
done
