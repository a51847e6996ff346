#!/usr/bin/env python3
"""Developer helper: write the dominant-kernel agreement note from one rocprofv3 kernel trace and two bench.py lines.

usage: tools/agreement.py <kernel_trace.csv> <bench_under_rocprofv3.json> <bench_default.json> <out.txt> <tag>

The note puts side by side, for the launch bench.py's `roofline` object reports, the hipEvent average measured inside
bench.py and the average rocprofv3 --kernel-trace measured for the same launches of the same command.
"""
import csv
import json
import sys


def main():
    trace, jprof, jdef, out, tag = sys.argv[1:6]
    prof = json.load(open(jprof))
    dflt = json.load(open(jdef))
    rows = list(csv.DictReader(open(trace)))
    for r in rows:
        r["_d"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        r["_s"] = int(r["Start_Timestamp"])
    rows.sort(key=lambda r: r["_s"])

    def named(sub):
        return [r for r in rows if sub in r["Kernel_Name"]]

    def avg(v):
        return sum(v) / max(len(v), 1)

    lines = []
    roof = prof["roofline"]
    lines.append("Launch bench.py's `roofline` reports (%s): %s; best launch of its family: %s" % (
        roof.get("selected_by", "largest summed time per step of its family"), roof["kernel"],
        (roof.get("family_best") or {}).get("kernel")))
    lines.append("command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps %d --warmup %d --repeats 5   "
                 "(profiles/%s_rocprofv3_kernel_stats.csv, profiles/%s_bench_under_rocprofv3.json)"
                 % (prof["steps"], prof["warmup"], tag, tag))
    # the trace rows of the reported launch: the kernel's instantiation(s) by name, on the largest grid they run on (= the
    # level-0 layer; the only level-0 layer in the fine form is dconv1 since round 4).  fp32 instantiations only: the bf16
    # networks of the line's `also` object launch other kernels.
    norm = lambda k: k.replace("(bool)0", "false").replace("(bool)1", "true")
    kname = roof["kernel"].split("/", 1)[1]
    pick = {"conv_bwd_logits_deep_kernel": lambda k: "conv_bwd_logits_deep_kernel" in k,
            "conv_w8_kernel<fwd>": lambda k: "conv_w8_kernel<false, true, 16, false" in norm(k),
            "conv_w8_kernel<data>": lambda k: "conv_w8_kernel<true, true, 16, false" in norm(k),
            "mlp_bwd_kernel<w>": lambda k: "mlp_bwd_w_split_kernel" in k,
            "mlp_bwd_kernel<dx>": lambda k: "mlp_bwd_dx_split_kernel" in k,
            "mlp_fwd_kernel": lambda k: "mlp_fwd_split_kernel" in k}.get(kname, lambda k: kname.split("<")[0] in k)
    sel = [r for r in rows if pick(r["Kernel_Name"])]
    grid = lambda r: int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    if sel:
        gmax = max(grid(r) for r in sel)
        d1 = [r["_d"] for r in sel if grid(r) == gmax]
        lines.append("  hipEvents inside bench.py (fgc_profile_*, %d steps), rocprofv3 attached      avg %.2f us"
                     % (prof["steps"], roof["avg_kernel_us"]))
        lines.append("  rocprofv3 --kernel-trace, same command, the %d launches of that kernel on its largest grid   avg %.2f us (min %.2f, max %.2f)"
                     % (len(d1), avg(d1), min(d1), max(d1)))
        lines.append("  hipEvents, no profiler, default arguments (profiles/%s_bench.json)        avg %.2f us"
                     % (tag, dflt["roofline"]["avg_kernel_us"]))
    else:
        lines.append("  (no trace rows matched %s)" % kname)
    kern = prof.get("kernels", {})
    other = []
    for sub, label, key in (("mlp_bwd_kernel", "mlp_bwd_kernel", "bwd:mlp/mlp_bwd_kernel"),
                            ("mlp_fwd_split_kernel", "mlp_fwd_split_kernel", "fwd:mlp/mlp_fwd_kernel")):
        v = [r["_d"] for r in named(sub)]
        if v:
            other.append("%s avg %.2f us over %d launches (hipEvents %s)"
                         % (label, avg(v), len(v), "%.2f" % kern[key]["avg_us"] if key in kern else "-"))
    dl = named("conv_bwd_logits_deep_kernel")
    if dl:
        gmax = max(int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]) for r in dl)
        v = [r["_d"] for r in dl if (int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])) == gmax]
        key = "bwd:dconv1/conv_bwd_logits_deep_kernel"
        other.append("conv_bwd_logits_deep_kernel (level 0, both layers) avg %.2f us over %d (hipEvents dconv1 %s)"
                     % (avg(v), len(v), "%.2f" % kern[key]["avg_us"] if key in kern else "-"))
    lines.append("other families, same trace: " + "; ".join(other))
    lines.append("families (bench.py, profiler attached):")
    for f in prof.get("families", []):
        lines.append("  " + json.dumps(f))
    lines.append("families (bench.py, no profiler, default arguments: %d steps after %d warm-up steps; profiles/%s_bench.json):"
                 % (dflt["steps"], dflt["warmup"], tag))
    for f in dflt.get("families", []):
        lines.append("  " + json.dumps(f))
    lines.append("step: %.4f ms (rocprofv3 attached, --steps %d --warmup %d), %.4f ms without (default arguments); value %.1f / %.1f facets/s"
                 % (prof["ms_per_step"], prof["steps"], prof["warmup"], dflt["ms_per_step"], prof["value"], dflt["value"]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main()
