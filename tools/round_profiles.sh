#!/bin/bash
# usage: tools/round_profiles.sh <outdir> <round-tag e.g. r02>      (GPU box)
# Regenerates every measurement DESIGN.md quotes: the default bench line, the driver-style short line,
# the rocprofv3 kernel statistics + steady-state timeline of that same command, the two PMC passes of
# the pipeline (per-kernel L2<->fabric traffic), the PMC passes of the row gather in isolation
# (calibrated), the sampling-only timeline, the model-step kernel table and the S-products line.
# PARTS="1 2 3 4" (default all) runs a subset: 1 = bench lines + kernel statistics / timeline, 2 = PMC passes,
# 3 = sampling only + model-step tables, 4 = the other workloads and legs  (a gpurun call is limited to 20 minutes)
out=$1; tag=$2
PARTS=${PARTS:-"1 2 3 4"}
want() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
set -x
if want 1; then
timeout -k 10 400 python3 bench.py > "$out/${tag}_bench_papers.json" 2> "$out/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-model-step --no-cpu-baseline > "$out/${tag}_bench_papers_steps20.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --workload S-products --no-cpu-baseline --no-model-step > "$out/${tag}_bench_products.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --force-distributed --no-cpu-baseline --no-model-step > "$out/${tag}_bench_papers_force_distributed.json" 2>> "$out/bench.err" || exit 1
# kernel statistics + timeline of the default command (model step, CPU leg and the whole-epoch legs off: the statistics
# then cover priming + warm-up + the timed windows, the launches the line's live HIP-event timing samples)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "${tag}_bench_papers" -- python3 bench.py --no-cpu-baseline --no-model-step --epochs 0 > "$out/prof_bench.json" 2>> "$out/bench.err" || exit 1
f=$(find "$out" -name "${tag}_bench_papers_kernel_trace.csv" | head -1)
python3 tools/trace_report.py "$f" 192 > "$out/${tag}_pipeline_trace_report.txt"
rm -f "$f" "$out"/${tag}_bench_papers_agent_info.csv "$out"/${tag}_bench_papers_domain_stats.csv
(head -1 "$out/${tag}_bench_papers_kernel_stats.csv"; grep "spp::" "$out/${tag}_bench_papers_kernel_stats.csv") > "$out/k.tmp" && mv "$out/k.tmp" "$out/${tag}_bench_papers_kernel_stats.csv"
fi
if want 2; then
# PMC: pipeline traffic per kernel (single-GPU path, then the partitioned path at world size 1)
tools/pmc_pipeline.sh "$out" "${tag}_pipeline_pmc" > /dev/null || exit 1
EXTRA_ARGS="--gpus 1 --force-distributed" tools/pmc_pipeline.sh "$out" "${tag}_partitioned_pmc" > /dev/null || exit 1
# PMC: the row gather in isolation, papers shape, with its calibration launch
for c in FETCH_SIZE WRITE_SIZE; do
  F=128 STRIDE=256 TABLE_ROWS=111059956 ROWS=947000 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out" -o "g_$c" -- python3 tools/pmc_gather.py > /dev/null 2>> "$out/bench.err" || exit 1
done
python3 tools/pmc_gather_report.py "$out/g_FETCH_SIZE_counter_collection.csv" "$out/g_WRITE_SIZE_counter_collection.csv" 256 "$out/${tag}_deliver_pmc_papers.json" 256 947000 111059956
for c in FETCH_SIZE WRITE_SIZE; do (head -1 "$out/g_${c}_counter_collection.csv"; grep "spp::" "$out/g_${c}_counter_collection.csv") > "$out/${tag}_gather_pmc_papers_${c}.csv"; rm -f "$out"/g_${c}_*; done
fi
if want 3; then
# sampling only
CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o chain -- python3 tools/microbench.py chain > "$out/${tag}_chain_only.log" 2>&1 || exit 1
f=$(find "$out" -name "chain_kernel_trace.csv" | head -1); python3 tools/trace_report.py "$f" 256 ${CHAIN_GROUP:-16} > "$out/${tag}_chain_only_trace_report.txt"; rm -f "$f" "$out"/chain_agent_info.csv
grep "chain only" "$out/${tag}_chain_only.log" >> "$out/${tag}_chain_only_trace_report.txt"
# model step
for m in sage gat; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "m_$m" -- python3 tools/model_step_profile.py $m 30 > "$out/m_$m.log" 2>&1 || exit 1
  (grep MODEL_STEP "$out/m_$m.log"; python3 tools/kstats.py "$out/m_${m}_kernel_stats.csv" 35 35 | grep -E "^ +[0-9.]+ us/step +[0-9]+\.0/step") > "$out/${tag}_model_step_${m}_kernel_stats.txt"
  rm -f "$out"/m_${m}_*
done
fi
set +x
ls -la "$out"
if want 4; then
# the other workloads and legs DESIGN.md quotes (one line each)
set -x
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/${tag}_bench_papers_driver_command.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --workload S-arxiv --no-cpu-baseline --no-model-step > "$out/${tag}_bench_arxiv.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --workload S-mag --no-cpu-baseline --no-model-step > "$out/${tag}_bench_mag.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 400 python3 bench.py --model gat --no-cpu-baseline > "$out/${tag}_bench_papers_gat.json" 2>> "$out/bench.err" || exit 1
timeout -k 10 400 python3 bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline > "$out/${tag}_bench_papers_force_distributed_ddp.json" 2>> "$out/bench.err" || exit 1
SPP_GROUP_DELIVERY=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-model-step > "$out/${tag}_bench_papers_group_delivery.json" 2>> "$out/bench.err" || exit 1
SPP_GROUP_DELIVERY=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step > "$out/${tag}_bench_papers_steps20_group_delivery.json" 2>> "$out/bench.err" || exit 1
fi
set +x
ls -la "$out"
