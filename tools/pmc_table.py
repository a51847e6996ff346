"""Developer helper: per-kernel PMC table from rocprofv3's rocpd database(s) (level-0 conv launches = largest grid)."""
import sqlite3, sys, collections

def table(path):
    c = sqlite3.connect(path)
    rows = c.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, vgpr_count, accum_vgpr_count, "
                     "lds_block_size, counter_name, value, duration from counters_collection").fetchall()
    disp = collections.OrderedDict()
    for did, kn, gs, wg, vg, ag, lds, cn, val, dur in rows:
        d = disp.setdefault(did, {"k": kn, "grid": gs, "wg": wg, "vgpr": vg, "agpr": ag, "lds": lds, "dur": dur, "c": {}})
        d["c"][cn] = d["c"].get(cn, 0.0) + val
    return disp

def short(k):
    k = k.replace("void ", "").replace("fgc::", "")
    return k.split("(")[0][:48]

if __name__ == "__main__":
    for p in sys.argv[1:]:
        disp = table(p)
        # per kernel name: keep the dispatch with the largest grid (level-0 instance), last occurrence
        best = {}
        for did, d in disp.items():
            key = short(d["k"])
            if "fgc" not in d["k"] and "conv" not in d["k"] and "mlp" not in d["k"] and "gemm" not in d["k"]:
                continue
            if key not in best or d["grid"] >= best[key]["grid"]:
                best[key] = d
        names = sorted({n for d in best.values() for n in d["c"]})
        print("#", p)
        print("%-48s %9s %5s %4s %4s %6s %8s " % ("kernel", "grid", "wg", "vgpr", "agpr", "lds", "dur_us") + " ".join("%14s" % n[-14:] for n in names))
        for key, d in sorted(best.items(), key=lambda kv: -kv[1]["dur"]):
            print("%-48s %9d %5d %4d %4d %6d %8.1f " % (key, d["grid"], d["wg"], d["vgpr"], d["agpr"], d["lds"], d["dur"] / 1e3) +
                  " ".join("%14.4g" % d["c"].get(n, float("nan")) for n in names))
