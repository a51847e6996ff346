#!/bin/bash
# Developer helper: alternating same-box runs of bench.py with and without an environment switch (library options are read
# from FGC_<NAME> at start-up), ms per step and the kernel lines that match a pattern.
# usage: tools/ab_env.sh <pattern> <rounds> VAR=value [bench args...]
pat=$1; rounds=$2; sw=$3; shift 3
for i in $(seq 1 $rounds); do
  for mode in off on; do
    if [ $mode = on ]; then export "$sw"; else unset "${sw%%=*}"; fi
    FGC_BENCH_NO_ALSO=1 python bench.py --no-cpu-baseline --dump-kernels /tmp/k.txt "$@" > /tmp/b.json 2>/tmp/b.err || { echo "$mode FAILED"; tail -3 /tmp/b.err; continue; }
    python - "$mode $sw" "$pat" <<'PY'
import json, sys
j = json.load(open("/tmp/b.json"))
lines = [l for l in open("/tmp/k.txt") if sys.argv[2] in l]
print("%-32s %.4f ms/step  loss %.4f" % (sys.argv[1], j["ms_per_step"], j["loss_deg"]))
for l in lines: print("      " + l.rstrip()[:150])
PY
  done
done
