#!/bin/bash
# usage: tools/pmc_pipeline.sh <outdir> <tag> [ENV=VAL ...]   (GPU box)
# Two rocprofv3 PMC passes over a short bench.py run (FETCH_SIZE, then WRITE_SIZE: separate passes, no
# tracing domains) and the per-kernel L2<->fabric traffic table per batch (tools/pmc_pipeline_report.py).
out=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
ARGS="--steps 48 --warmup 8 --prime 8 --no-cpu-baseline --no-model-step $EXTRA_ARGS"   # EXTRA_ARGS="--gpus 1 --force-distributed": the partitioned path
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out" -o "${tag}_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$out/${tag}_fetch.err" || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out" -o "${tag}_write" -- python3 bench.py $ARGS > /dev/null 2> "$out/${tag}_write.err" || exit 1
f=$(find "$out" -name "${tag}_fetch_counter_collection.csv" | head -1)
w=$(find "$out" -name "${tag}_write_counter_collection.csv" | head -1)
# batches delivered = k_deliver launches (+ G per group launch); batches sampled = k_seed_init launches x G (G = batches per sampling launch)
G=${SPP_GROUP_SIZE:-16}; nd=$(( $(grep -c "k_deliver<" "$f") + G * $(grep -c "k_deliver_group" "$f") )); ns=$(( $(grep -c "k_seed_init" "$f") * G ))
python3 tools/pmc_pipeline_report.py "$f" "$w" "$nd" "$ns" > "$out/${tag}_traffic_per_kernel.txt"
# keep the spp:: rows only (the torch graph construction dominates the raw files)
for x in "$f" "$w"; do (head -1 "$x"; grep "spp::" "$x") > "$x.tmp" && mv "$x.tmp" "$x"; done
rm -f "$out"/${tag}_*_agent_info.csv
cat "$out/${tag}_traffic_per_kernel.txt"
