#!/bin/bash
# Developer helper: time bench.py with alternative builds of libfgc.so (FGC_LIB), printing ms/step and one kernel line.
# usage: tools/bench_variants.sh <kernel-substring> lib1.so lib2.so ...
pat=$1; shift
for lib in "$@"; do
  FGC_LIB=$lib python bench.py --no-cpu-baseline --dump-kernels /tmp/k.txt > /tmp/b.json 2>/tmp/b.err || { echo "$lib FAILED"; tail -3 /tmp/b.err; continue; }
  python - "$lib" "$pat" <<'PY'
import json, sys
j = json.load(open("/tmp/b.json"))
lines = [l for l in open("/tmp/k.txt") if sys.argv[2] in l]
print("%-48s %.3f ms/step  loss %.4f | %s" % (sys.argv[1].split("/")[-1], j["ms_per_step"], j["loss_deg"], lines[0].split("avg")[1].split("per-step")[0].strip() if lines else "-"))
PY
done
