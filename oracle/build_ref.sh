#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
#
# Builds the UNMODIFIED reference native module (pybind11/torch extension
# `fast_sampler`) straight from its sources where they lie under
# /root/reference/fast_sampler, into oracle/_ref/fast_sampler.so.
#
#  * no reference source is copied or patched; nothing is written outside
#    oracle/_ref/ (git-ignored, but it travels to the GPU box with gpurun);
#  * the reference's own build system (setup.py) is NOT run: this is a direct
#    two-file clang++ invocation;
#  * clang (ROCm LLVM) is used because g++ 11 rejects the nested AT_DISPATCH in
#    full_sample (fast_sampler.cpp:355-365) with torch 2.10 headers;
#  * libomp is linked explicitly (the reference's setup.py forgets to).
#
# The resulting module is used (a) to generate tests/golden/* (see
# tests/golden/make_golden.py) and thereby pin oracle/spp_oracle.c, and (b) as
# the `cpu_baseline.kind == "reference"` leg of bench.py.
set -euo pipefail
REF=${SPP_REFERENCE_DIR:-/root/reference/fast_sampler}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ ! -d "$REF" ]; then
  echo "[build_ref] $REF not present (GPU box?) - keeping prebuilt $OUT/fast_sampler.so if any"
  exit 0
fi
if [ -f "$OUT/fast_sampler.so" ] && [ "$OUT/fast_sampler.so" -nt "$REF/fast_sampler.cpp" ] \
   && [ "$OUT/fast_sampler.so" -nt "$HERE/build_ref.sh" ]; then
  echo "[build_ref] up to date"
  exit 0
fi
LLVM=/opt/rocm/lib/llvm
PY=${PYTHON:-python3}
TORCH_DIR=$($PY -c "import torch, os; print(os.path.dirname(torch.__file__))")
PYINC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
ABI=$($PY -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
CXX="$LLVM/bin/clang++"
FLAGS=(-O3 -march=x86-64-v3 -std=c++17 -fPIC -fopenmp -DAT_PARALLEL_OPENMP -DNDEBUG
       -DTORCH_EXTENSION_NAME=fast_sampler -DTORCH_API_INCLUDE_EXTENSION_H
       -D_GLIBCXX_USE_CXX11_ABI=$ABI -w
       -I"$REF" -I"$REF/parallel-hashmap"
       -isystem "$TORCH_DIR/include" -isystem "$TORCH_DIR/include/torch/csrc/api/include"
       -isystem "$PYINC")
"$CXX" "${FLAGS[@]}" -c "$REF/fast_sampler.cpp" -o "$OUT/fast_sampler.o" &
"$CXX" "${FLAGS[@]}" -c "$REF/range_partition_book.cpp" -o "$OUT/range_partition_book.o" &
wait
"$CXX" -shared "$OUT/fast_sampler.o" "$OUT/range_partition_book.o" -o "$OUT/fast_sampler.so" \
  -L"$TORCH_DIR/lib" -ltorch -ltorch_cpu -ltorch_python -lc10 \
  -L"$LLVM/lib" -lomp -lpthread \
  -Wl,-rpath,"$TORCH_DIR/lib" -Wl,-rpath,"$LLVM/lib"
rm -f "$OUT/fast_sampler.o" "$OUT/range_partition_book.o"
echo "[build_ref] built $OUT/fast_sampler.so"
