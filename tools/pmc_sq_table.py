"""Developer helper: tables of the counter passes of tools/pmc_sq.sh (last step only: the largest-grid launch per kernel name
and the level-0 launches of the conv kernels)."""
import csv, glob, os, sys, collections

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print("# %s: no counters collected" % d)
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "fgc::" not in k:
            continue
        e = disp.setdefault(int(r["Dispatch_Id"]), {"k": k, "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]),
                                                    "lds": int(r["LDS_Block_Size"]),
                                                    "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    best = collections.OrderedDict()
    for did, e in disp.items():
        name = e["k"].replace("void ", "").replace("fgc::", "").split("(")[0][:52]
        if name not in best or e["grid"] >= best[name]["grid"]:
            best[name] = e
    names = sorted({n for e in best.values() for n in e["c"]})
    print("# pass %s" % os.path.basename(d))
    print("%-52s %9s %4s %6s %8s " % ("kernel (largest-grid launch, last step)", "grid", "vgpr", "lds", "us") +
          " ".join("%13s" % n.replace("SQ_", "")[-13:] for n in names))
    for name, e in sorted(best.items(), key=lambda kv: -kv[1]["us"]):
        if e["us"] < 15:
            continue
        print("%-52s %9d %4d %6d %8.1f " % (name, e["grid"], e["vgpr"], e["lds"], e["us"]) +
              " ".join("%13.4g" % e["c"].get(n, float("nan")) for n in names))
    print()
