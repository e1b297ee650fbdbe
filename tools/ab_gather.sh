#!/bin/bash
# usage: tools/ab_gather.sh <outfile> <rounds> v1 v2 ...  (GPU box): the row gather alone (tools/microbench.py gather) under source variants (.abt/<name>/)
out=$1; rounds=$2; shift 2
mkdir -p "$(dirname "$out")"
keep=$(mktemp -d /tmp/abt_keep.XXXXXX)
cp salient_plusplus_amd/csrc/*.hip salient_plusplus_amd/csrc/*.cuh salient_plusplus_amd/csrc/*.h "$keep"/
restore() { cp "$keep"/* salient_plusplus_amd/csrc/; python3 -m salient_plusplus_amd.build > /dev/null 2>&1; }
trap restore EXIT
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp "$keep"/* salient_plusplus_amd/csrc/
    [ -d ".abt/$v" ] && cp .abt/$v/* salient_plusplus_amd/csrc/
    python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build of $v failed" >> "$out"; continue; }
    for w in ${WGS:-16}; do
      echo "round $r variant $v wg/cu $w: $(SPP_GATHER_WG_PER_CU=$w WL=S-papers timeout -k 10 200 python3 tools/microbench.py gather 2>/dev/null | grep 'U=770000 F=128 idx=i32' | tr '\n' ' ')" >> "$out"
    done
  done
done
cat "$out"
