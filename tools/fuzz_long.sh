#!/bin/bash
# usage: tools/fuzz_long.sh <outdir> [rounds=5] [examples per round=2000] [exchange examples per round=500]   (GPU box)
# The hypothesis suites (single-GPU Session, partitioned exchange on in-process ranks, the row gather over random widths /
# strides / misaligned addresses) with many freshly drawn cases, every batch bit for bit against the oracle.  A failure prints the falsifying example; a hang or crash leaves the case
# that was running as the last line of <outdir>/cases_*.log (SPP_FUZZ_LOG).
out=$1; rounds=${2:-5}; n=${3:-2000}; m=${4:-500}
mkdir -p "$out"
for r in $(seq 1 $rounds); do
  : > "$out/cases_session.log"; : > "$out/cases_exchange.log"
  SPP_FUZZ_LOG="$out/cases_session.log" SPP_FUZZ_RANDOM=1 SPP_FUZZ_EXAMPLES=$n timeout -k 10 300 python3 -m pytest tests/test_gpu_random_graphs.py -x -q -m gpu \
    -p no:cacheprovider -k "test_random_graph_against_the_oracle or free_functions or lookups" --hypothesis-show-statistics > "$out/fuzz_session_$r.txt" 2>&1 || { tail -40 "$out/fuzz_session_$r.txt"; tail -1 "$out/cases_session.log"; exit 1; }
  echo "round $r session, free functions, lookups: $(grep -o '[0-9]* passing' "$out/fuzz_session_$r.txt" | tr '\n' ' ')"
  SPP_FUZZ_LOG="$out/cases_exchange.log" SPP_FUZZ_RANDOM=1 SPP_FUZZ_EXAMPLES=$m timeout -k 10 300 python3 -m pytest tests/test_gpu_random_exchange.py -x -q -m gpu \
    -p no:cacheprovider --hypothesis-show-statistics > "$out/fuzz_exchange_$r.txt" 2>&1 || { tail -60 "$out/fuzz_exchange_$r.txt"; tail -1 "$out/cases_exchange.log"; exit 1; }
  SPP_FUZZ_RANDOM=1 SPP_FUZZ_EXAMPLES=$m timeout -k 10 600 python3 -m pytest tests/test_gpu_random_graphs.py -x -q -m gpu -p no:cacheprovider \
    -k test_random_facade_configurations --hypothesis-show-statistics > "$out/fuzz_facade_$r.txt" 2>&1 || { tail -60 "$out/fuzz_facade_$r.txt"; exit 1; }
  echo "round $r facade: $(grep -o '[0-9]* passing' "$out/fuzz_facade_$r.txt")"
  echo "round $r exchange: $(grep -o '[0-9]* passing' "$out/fuzz_exchange_$r.txt")"
  SPP_FUZZ_RANDOM=1 SPP_FUZZ_EXAMPLES=$n timeout -k 10 300 python3 -m pytest tests/test_gpu_random_kernels.py -x -q -m gpu -p no:cacheprovider \
    -k test_gather_rows_random_shapes --hypothesis-show-statistics > "$out/fuzz_gather_$r.txt" 2>&1 || { tail -40 "$out/fuzz_gather_$r.txt"; exit 1; }
  echo "round $r row gather: $(grep -o '[0-9]* passing' "$out/fuzz_gather_$r.txt")"
done
