#!/bin/bash
# Developer helper: per-kernel times of bench.py under environment switches.
# usage: tools/env_bench.sh "<kernel-prefix> ..." "VAR=val VAR2=val" "VAR=val" ...   ("-" = no switch)
pats=$1; shift
for envs in "$@"; do
  if [ "$envs" = "-" ]; then e=""; else e="$envs"; fi
  env $e timeout -k 10 200 python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --dump-kernels /tmp/k.txt > /tmp/b.json 2>/tmp/b.err || { echo "$envs FAILED"; tail -3 /tmp/b.err; continue; }
  python - "$envs" $pats <<'PY'
import json, sys
j = json.load(open("/tmp/b.json"))
out = []
for pat in sys.argv[2:]:
    for l in open("/tmp/k.txt"):
        if l.startswith(pat):
            out.append("%s %s" % (pat, l.split("avg")[1].split("us")[0].strip()))
print("%-40s %.3f ms/step | " % (sys.argv[1], j["ms_per_step"]) + "  ".join(out))
PY
done
