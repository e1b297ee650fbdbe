#!/bin/bash
# round 5: K = 20 windows under stream priorities (the closing synchronize waits for the chain in flight)
OUT=${1:-gpurun_out/r5d}; mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" SPP_BENCH_TAIL=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $OUT/$tag.err $tag <<'PY'
import json,sys,re
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tails=[int(m.group(1)) for m in re.finditer(r"chains in flight\) \+(\d+) us", open(sys.argv[2]).read())]
print(f"{sys.argv[3]:28s} ms/step {d['ms_per_step']:.4f} deliver_us {1e3*d['roofline']['avg_launch_ms']:.1f} mean tail {sum(tails)/max(1,len(tails)):.0f} us  windows {d['windows']['ms_per_step_all']}")
PY
}
for rep in 1 2; do
run base_$rep X=1
run samp_high_$rep SPP_SAMPLING_PRIORITY=high
run deliv_low_$rep SPP_DELIVERY_PRIORITY=low
run streams1_$rep SPP_WORK_STREAMS=1
run streams3_$rep SPP_WORK_STREAMS=3
done
