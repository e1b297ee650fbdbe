#!/bin/bash
# round 5: the tests touched so far
OUT=${1:-gpurun_out/r5g}; mkdir -p $OUT
python -m pytest tests/test_gpu_native_exchange.py -x -q -k "counting" > $OUT/t1.txt 2>&1; tail -3 $OUT/t1.txt
python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_api_misc.py tests/test_gpu_pipeline.py -x -q > $OUT/t2.txt 2>&1; tail -3 $OUT/t2.txt
