#!/bin/bash
# round 5: the delivery stream confined to a CU mask UNDER THE MODEL STEP (never tried there: the GEMM phases leave slack)
OUT=${1:-gpurun_out/r5f}; mkdir -p $OUT
run() { tag=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-leg > $OUT/$tag.json 2> $OUT/$tag.err || { tail -5 $OUT/$tag.err; return; }
  python - $OUT/$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); m=d["model_step"]
print(f"{sys.argv[2]:24s} data-path ms/step {d['ms_per_step']:.4f} deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f} | model only {m['ms_per_step_model_only_resident_batch']:.4f} with data {m['ms_per_step_with_data_path']:.4f} cost {m['ms_per_step_with_data_path']-m['ms_per_step_model_only_resident_batch']:.4f}")
PY
}
run base X=1
run mask32 SPP_DELIVERY_CU_MASK=32
run mask64 SPP_DELIVERY_CU_MASK=64
run mask128 SPP_DELIVERY_CU_MASK=128
run mask64_blocked SPP_DELIVERY_CU_MASK=64 SPP_CU_MASK_LAYOUT=blocked
run mask64_both SPP_DELIVERY_CU_MASK=64 SPP_SAMPLING_CU_MASK=64
run base2 X=1
