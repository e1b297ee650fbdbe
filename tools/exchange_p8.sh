#!/bin/bash
# usage: [WL=S-papers NB=32 EPOCHS=1] tools/exchange_p8.sh <outdir> <tag e.g. r04_exchange_p8>     (GPU box)
# Kernel trace + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, no tracing domains) of the 8-rank exchange
# rehearsal (tools/exchange_p8.py) and the per-kernel table (tools/exchange_p8_report.py).
out=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
export VERIFY=0 SPP_ALLOW_LOCAL_COMM=1
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$out" -o "${tag}_kt" -- python3 tools/exchange_p8.py 8 ${NB:-24} ${EPOCHS:-2} > "$out/${tag}.log" 2> "$out/${tag}.err" || { tail -5 "$out/${tag}.err"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d "$out" -o "${tag}_$c" -- python3 tools/exchange_p8.py 8 ${NB:-24} ${EPOCHS:-2} > /dev/null 2>> "$out/${tag}.err" || { tail -5 "$out/${tag}.err"; exit 1; }
done
t=$(find "$out" -name "${tag}_kt_kernel_trace.csv" | head -1)
f=$(find "$out" -name "${tag}_FETCH_SIZE_counter_collection.csv" | head -1)
w=$(find "$out" -name "${tag}_WRITE_SIZE_counter_collection.csv" | head -1)
python3 tools/exchange_p8_report.py "$t" "$f" "$w" "$out/${tag}.log" > "$out/${tag}_per_kernel.txt" || exit 1
(grep "^rank \|^EXCHANGE_P8" "$out/${tag}.log") > "$out/${tag}_composition.txt"
for x in "$f" "$w"; do (head -1 "$x"; grep "spp::" "$x" | grep "k_serve_rows\|k_deliver\|k_pack_remote\|k_gpart") > "$out/$(basename "$x" .csv)_spp.csv"; rm -f "$x"; done
rm -f "$t" "$out"/${tag}_*_agent_info.csv
cat "$out/${tag}_per_kernel.txt"
