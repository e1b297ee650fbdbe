#!/bin/bash
# round 5: k_bucket_dedup with its first pairs loaded in the first round trip (prebuilt variants under .abt/variants)
OUT=${1:-gpurun_out/r5l}; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_random_graphs.py -x -q > $OUT/tests.txt 2>&1 || { tail -20 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
bash tools/ab_libs.sh $OUT/ab_k192.txt 3 "--steps 192 --warmup 5" .abt/variants/base_pairs.so .abt/variants/spec_pairs.so
bash tools/ab_libs.sh $OUT/ab_k20.txt 3 "--steps 20 --warmup 5" .abt/variants/base_pairs.so .abt/variants/spec_pairs.so
