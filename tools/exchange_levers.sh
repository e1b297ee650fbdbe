#!/bin/bash
# round 5: the multi-GPU levers rehearsed with eight in-process ranks on one GPU at S-papers: cache fraction 0.1 / 0.2 / 0.4
# (fetched fraction, exchange bytes per batch and rank); composition only -- ranks share the GPU, times mean nothing
OUT=${1:-gpurun_out/r5m}; mkdir -p $OUT
for cf in 0.1 0.2 0.4; do
  WL=S-papers CACHE_FRAC=$cf VERIFY=${VERIFY:-1} timeout -k 10 900 python tools/exchange_p8.py 8 ${NB:-16} 1 > $OUT/exchange_p8_papers_cache$cf.log 2> $OUT/exchange_p8_papers_cache$cf.err || { tail -5 $OUT/exchange_p8_papers_cache$cf.err; exit 1; }
  grep "^rank 0\|^EXCHANGE_P8" $OUT/exchange_p8_papers_cache$cf.log
done
