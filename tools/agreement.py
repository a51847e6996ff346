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
    lines.append("Reported family of bench.py (largest summed time per step): %s; reported through its launch with the most "
                 "algorithmic work: %s" % (roof["family"], roof["kernel"]))
    lines.append("command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps %d --warmup %d --repeats 5   "
                 "(profiles/%s_rocprofv3_kernel_stats.csv, profiles/%s_bench_under_rocprofv3.json)"
                 % (prof["steps"], prof["warmup"], tag, tag))
    # the forward conv launches of level 0: the fp32 forward instantiation (<DATA = false, FAST, QS = 16, BF = false, ...>; the
    # bf16 networks of the line's `also` object launch the BF = true one on the same grid) on its largest grid = dconv1 (upconv1
    # runs in the pair form since round 4)
    fwd = [r for r in rows if "conv_w8_kernel<false, true, 16, false" in r["Kernel_Name"].replace("(bool)0", "false").replace("(bool)1", "true")]
    if not fwd:
        fwd = [r for r in named("conv_w8_kernel<false")]
    if fwd:
        gmax = max(int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]) for r in fwd)
        big = [r for r in fwd if (int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])) == gmax]
        d1 = [r["_d"] for r in big]
        u1 = []
        lines.append("  hipEvents inside bench.py (fgc_profile_*, %d steps), rocprofv3 attached      avg %.2f us"
                     % (prof["steps"], roof["avg_kernel_us"]))
        lines.append("  rocprofv3 --kernel-trace, same command, the %d dconv1 forward launches     avg %.2f us (min %.2f, max %.2f)"
                     % (len(d1), avg(d1), min(d1), max(d1)))
        if u1:
            lines.append("  (upconv1 forward, same grid: avg %.2f us over %d)" % (avg(u1), len(u1)))
        lines.append("  hipEvents, no profiler, default arguments (profiles/%s_bench.json)        avg %.2f us"
                     % (tag, dflt["roofline"]["avg_kernel_us"]))
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
