#!/bin/bash
# round 5: the span form of the row gather (200-byte rows, 16-byte accesses) -- tests, lone gather and S-products A/B
set -e
OUT=${1:-gpurun_out/r5a}
mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -k "gather" > $OUT/test_gather.txt 2>&1 || { tail -30 $OUT/test_gather.txt; exit 1; }
tail -3 $OUT/test_gather.txt
for span in 1 0; do
  SPP_GATHER_SPAN=$span python tools/microbench.py gather > $OUT/microbench_gather_span$span.txt 2>&1
  tail -4 $OUT/microbench_gather_span$span.txt
done
for rep in 1 2; do
  for span in 1 0; do
    SPP_GATHER_SPAN=$span python bench.py --workload S-products --no-model-step --no-cpu-baseline > $OUT/bench_products_span${span}_$rep.json 2> $OUT/bench_products_span${span}_$rep.err
    python - $OUT/bench_products_span${span}_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"])
PY
  done
done
