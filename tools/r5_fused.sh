#!/bin/bash
# round 5: the fused first layer (Session(table_features) + models.SAGE from the resident table): tests, then the bench legs
set -e
OUT=${1:-gpurun_out/r5e}; mkdir -p $OUT
python -m pytest tests/test_gpu_model_step.py -x -q -k "fused or table_features or sage_matches or end_to_end" > $OUT/test_fused.txt 2>&1 || { tail -40 $OUT/test_fused.txt; exit 1; }
tail -3 $OUT/test_fused.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_papers_fused.json 2> $OUT/bench_papers_fused.err || { tail -20 $OUT/bench_papers_fused.err; exit 1; }
python - $OUT/bench_papers_fused.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m=d["model_step"]
print("value", d["value"], "ms/step", d["ms_per_step"])
print("default: model only", m["ms_per_step_model_only_resident_batch"], "with data", m["ms_per_step_with_data_path"], "cost", m.get("data_path_cost_ms"))
f=m["fused_first_layer"]
print("fused:   model only", f["ms_per_step_model_only_resident_batch"], "with data", f["ms_per_step_with_data_path"], "cost", f["data_path_cost_ms"], "epoch", f["epoch_time_s_with_model_step"], "vs", d["epoch_time_s_with_model_step"])
PY
