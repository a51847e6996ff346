#!/bin/bash
# Developer helper (GPU box): regenerate every measurement file of a round at one commit.
# usage: tools/refresh_profiles.sh <tag> <commit> [first-stage] [round prefix, default r3]  ->  gpurun_out/refresh/<tag>_*  (copy what should be judged
# into profiles/).  Stages: 1 traffic, 2 fp32 rocprof + default bench + agreement note, 3 bf16, 4 configs c3 / c4 / c5.
# Steps: PMC traffic passes (fp32, bf16) -> profiles traffic files in place, so the bench lines after them carry this build's
# `traffic`; rocprofv3 --kernel-trace --stats of the bench command; the default bench line; bf16, c3, c4, c5 lines.
tag=$1; commit=${2:-unknown}; first=${3:-1}; rnd=${4:-r3}
o=gpurun_out/refresh; rm -rf $o; mkdir -p $o
export TMPDIR=/tmp
set -e
if [ $first -le 1 ]; then
bash tools/pmc_traffic.sh f32 f32 $commit > $o/traffic_f32.log 2>&1
cp gpurun_out/traffic_f32.json profiles/${rnd}_traffic_families.json; cp gpurun_out/traffic_f32.txt profiles/${rnd}_traffic_summary.txt
bash tools/pmc_traffic.sh bf16 bf16 $commit > $o/traffic_bf16.log 2>&1
cp gpurun_out/traffic_bf16.json profiles/${rnd}_traffic_families_bf16.json; cp gpurun_out/traffic_bf16.txt profiles/${rnd}_traffic_summary_bf16.txt
cp profiles/${rnd}_traffic_*.json profiles/${rnd}_traffic_summary*.txt $o/
echo "traffic done"
fi
if [ $first -le 2 ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python3 bench.py --steps 20 --warmup 5 --repeats 5 \
    > $o/${tag}_bench_under_rocprofv3.json 2> $o/prof.err
cp $(find $o/prof -name "*kernel_stats.csv" | head -1) $o/${tag}_rocprofv3_kernel_stats.csv
echo "rocprof done"
timeout -k 10 400 python3 bench.py > $o/${tag}_bench.json 2> $o/bench.err
python3 tools/agreement.py $(find $o/prof -name "*kernel_trace.csv" | head -1) $o/${tag}_bench_under_rocprofv3.json $o/${tag}_bench.json \
    $o/${tag}_dominant_kernel_agreement.txt $tag > /dev/null
echo "default bench done"
fi
if [ $first -le 3 ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_bf -o p -- python3 bench.py --dtype bf16 --steps 20 --warmup 5 --repeats 5 \
    --no-cpu-baseline > $o/bf16_bench_under_rocprofv3.json 2> $o/prof_bf.err
cp $(find $o/prof_bf -name "*kernel_stats.csv" | head -1) $o/bf16_rocprofv3_kernel_stats.csv
timeout -k 10 300 python3 bench.py --dtype bf16 --no-cpu-baseline > $o/bf16_bench.json 2> $o/bf16.err
echo "bf16 done"
fi
timeout -k 10 300 python3 bench.py --config c3 --no-cpu-baseline > $o/c3_bench.json 2> $o/c3.err
timeout -k 10 400 python3 bench.py --config c4 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $o/c4_bench.json 2> $o/c4.err
timeout -k 10 400 python3 bench.py --config c5 --no-cpu-baseline > $o/c5_bench.json 2> $o/c5.err
rm -rf $o/prof $o/prof_bf
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/refresh/*bench*.json")):
    try:
        d = json.load(open(f))
        print("%-60s %.4f ms/step  %.1f %s" % (f.split("/")[-1], d["ms_per_step"], d["value"], d["unit"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
