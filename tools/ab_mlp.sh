#!/bin/bash
# Developer helper: alternating same-box runs of bench.py with two builds of libfgc.so (FGC_LIB), ms per step and the
# kernel lines that match a pattern.   usage: tools/ab_mlp.sh <pattern> <rounds> libA.so libB.so
pat=$1; rounds=$2; shift 2
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    FGC_BENCH_NO_ALSO=1 FGC_LIB=$lib python bench.py --no-cpu-baseline --dump-kernels /tmp/k.txt > /tmp/b.json 2>/tmp/b.err || { echo "$lib FAILED"; tail -3 /tmp/b.err; continue; }
    python - "$lib" "$pat" <<'PY'
import json, sys
j = json.load(open("/tmp/b.json"))
lines = [l for l in open("/tmp/k.txt") if sys.argv[2] in l]
print("%-32s %.4f ms/step  loss %.4f" % (sys.argv[1].split("/")[-1], j["ms_per_step"], j["loss_deg"]))
for l in lines: print("      " + l.rstrip()[:150])
PY
  done
done
