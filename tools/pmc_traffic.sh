#!/bin/bash
# Developer helper (GPU box): whole-step and per-kernel HBM traffic from two separate rocprofv3 PMC passes.
# usage: tools/pmc_traffic.sh <tag> [f32|bf16] [commit]   ->  gpurun_out/traffic_<tag>.json / .txt
# (counters are collected with --kernel-trace only: no other tracing domain shares a pass with --pmc)
tag=$1; dt=${2:-f32}; commit=${3:-unknown}; steps=3
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 tools/pmc_steps.py $steps $dt > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 tools/pmc_steps.py $steps $dt > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
f=$(find $out/fetch -name "*counter_collection.csv" | head -1); w=$(find $out/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_collect.py $f $w $steps gpurun_out/traffic_$tag.json "$commit" "tools/pmc_traffic.sh $tag $dt" | tee gpurun_out/traffic_$tag.txt
