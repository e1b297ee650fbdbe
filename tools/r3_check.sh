#!/bin/bash
# usage: tools/r3_check.sh <outdir>   (GPU box): quick correctness subset + the driver-style bench line + the N>1 paths a 1-GPU box can run
out=$1; mkdir -p "$out"
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_ddp_interleave.py -m gpu -x -q > "$out/t.log" 2>&1; tail -3 "$out/t.log"
SPP_BENCH_STEP_TIMES=1 python bench.py --steps 20 --warmup 5 > "$out/driver_like.json" 2> "$out/driver_like.err"; echo "driver_like rc=$?"
python bench.py --no-cpu-baseline --no-model-step > "$out/k192.json" 2> /dev/null
python bench.py --gpus 2 > "$out/gpus2.out" 2>&1; echo "gpus2 rc=$?" >> "$out/gpus2.out"
python bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline > "$out/fd.json" 2> "$out/fd.err"; echo "fd rc=$?"
WL=S-papers python tools/microbench.py gather > "$out/mb.log" 2>&1; tail -8 "$out/mb.log"
python - "$out" <<'PY'
import json, sys
for f in ("driver_like.json", "k192.json", "fd.json"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{f}").read().strip().splitlines()[-1])
        m = d.get("model_step") or {}
        print(f, "ms_per_step", round(d["ms_per_step"], 4), "deliver_us", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), "frac", round(d["roofline"]["frac"], 3),
              "model", m.get("ms_per_step_model_only_resident_batch"), m.get("ms_per_step_with_data_path"), d.get("epoch_time_s_with_model_step"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
