#!/bin/bash
# Developer helper: per-kernel times of bench.py under alternative builds of libfgc.so (phase knock-outs, FGC_LIB).
# usage: tools/ko_bench.sh "<extra bench args>" "<kernel-substring> ..." lib1.so lib2.so ...   ("-" = the in-tree library)
extra=$1; shift
pats=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset FGC_LIB; else export FGC_LIB=$lib; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --dump-kernels /tmp/k.txt $extra > /tmp/b.json 2>/tmp/b.err || { echo "$lib FAILED"; tail -3 /tmp/b.err; continue; }
  python - "$lib" $pats <<'PY'
import json, sys
j = json.load(open("/tmp/b.json"))
out = []
for pat in sys.argv[2:]:
    for l in open("/tmp/k.txt"):
        if l.startswith(pat):
            out.append("%s %s" % (pat.split("/")[0] + ("<d>" if "data" in pat else ""), l.split("avg")[1].split("us")[0].strip()))
print("%-24s %.3f ms/step | " % (sys.argv[1].split("/")[-1], j["ms_per_step"]) + "  ".join(out))
PY
done
