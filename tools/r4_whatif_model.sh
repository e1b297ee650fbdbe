#!/bin/bash
# usage: tools/r4_whatif_model.sh <outfile> "<ENV=VAL ...>" ...     (GPU box)
# ms/step of the SAGE step fed by the data path (tools/overlap_trace.py, LEG=data, no profiler) under each environment
# given (one quoted argument per variant; "" = defaults), bracketed by the resident-batch leg.
out=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$(dirname "$out")"
: > "$out"
run() {  # leg, env string
  local line
  line=$(env $2 LEG=$1 timeout -k 10 300 python3 tools/overlap_trace.py ${ARCH:-sage} ${STEPS:-192} 2>/dev/null | grep OVERLAP_TRACE) || { echo "FAILED: $1 [$2]" | tee -a "$out"; return 1; }
  echo "$line   [$2]" | tee -a "$out"
}
run resident "" || exit 1
run rotate "" || exit 1
for v in "$@"; do run data "$v" || exit 1; done
run resident ""
run rotate ""
run data ""
