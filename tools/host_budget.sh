#!/bin/bash
# round 5: the host budget of one rank of an 8-rank node (16-core quota / 8 = 2 cores), rehearsed on one GPU:
# the partitioned path with the DDP model-step leg, confined to 2 / 4 cores and unconfined; then the single-GPU path
OUT=${1:-gpurun_out/r5h}; mkdir -p $OUT
for cores in 0 2 4; do
  python bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline --cpu-cores $cores > $OUT/host_budget_dist_cores$cores.json 2> $OUT/host_budget_dist_cores$cores.err || tail -5 $OUT/host_budget_dist_cores$cores.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cpu-cores $cores > $OUT/host_budget_single_cores$cores.json 2> $OUT/host_budget_single_cores$cores.err || tail -5 $OUT/host_budget_single_cores$cores.err
  python - $OUT/host_budget_dist_cores$cores.json $OUT/host_budget_single_cores$cores.json $cores <<'PY'
import json,sys
for f in sys.argv[1:3]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); m=d["model_step"]
    fl=m.get("fused_first_layer")
    print(f"cores {sys.argv[3]} {d['config']['parallelism'][:12]:12s} data path {d['ms_per_step']:.4f} (windows {min(d['windows']['ms_per_step_all']):.4f}-{max(d['windows']['ms_per_step_all']):.4f}) deliver_us {d['windows']['deliver_us_all']} | model only {m['ms_per_step_model_only_resident_batch']:.3f} with data {m['ms_per_step_with_data_path']:.3f}" + (f" | fused {fl['ms_per_step_model_only_resident_batch']:.3f} / {fl['ms_per_step_with_data_path']:.3f}" if fl else ""))
PY
done
