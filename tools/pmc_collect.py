"""Developer helper: HBM traffic per kernel launch and per whole step from two rocprofv3 PMC passes of tools/pmc_steps.py
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, both with --kernel-trace --output-format csv; counters never share
a pass with the tracing domains gpurun refuses).

    python tools/pmc_collect.py <fetch counter_collection.csv> <write counter_collection.csv> <steps> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section): hbm = 2 * FETCH + WRITE.  Launches are matched to layers by their order inside a
step (forward: conv2, conv3, dconv3, upconv2, dconv2, upconv1, dconv1; backward: the reverse, conv1 last)."""
import csv
import json
import sys

FWD = ["conv2", "conv3", "dconv3", "upconv2", "dconv2", "upconv1", "dconv1"]
BWD = ["dconv1", "upconv1", "dconv2", "upconv2", "dconv3", "conv3", "conv2"]


def load(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def short(k):
    k = k.replace("void ", "").replace("fgc::", "")
    return k.split("(")[0]


def label(names):
    """tag/kernel key (as bench.py prints them) for every dispatch of ONE step."""
    out, cnt = [], {}
    for k in names:
        s = short(k)
        c = cnt.get(s.split("<")[0], 0)
        key = None
        if s.startswith("conv_w8_kernel<false"):
            key = "fwd:%s/conv_w8_kernel<fwd>" % FWD[cnt.get("w8f", 0)]
            cnt["w8f"] = cnt.get("w8f", 0) + 1
        elif s.startswith("conv_w8_kernel<true"):
            key = "bwd:%s/conv_w8_kernel<data>" % BWD[cnt.get("w8d", 0)]
            cnt["w8d"] = cnt.get("w8d", 0) + 1
        elif s.startswith("conv_bwd_logits"):
            key = "bwd:%s/%s" % (BWD[cnt.get("k1", 0)], s.split("<")[0])
            cnt["k1"] = cnt.get("k1", 0) + 1
        elif s.startswith("gemm_tn"):
            i = cnt.get("tn", 0)
            key = "bwd:%s/gemm_tn_kernel:dW" % (BWD[i] if i < 7 else "conv1")
            cnt["tn"] = i + 1
        elif s.startswith("mlp_bwd_dx"):
            key = "bwd:mlp/mlp_bwd_kernel<dx>"
        elif s.startswith("mlp_bwd_w"):
            key = "bwd:mlp/mlp_bwd_kernel<w>"
        elif s.startswith("mlp_bwd") or s.startswith("mlp_fwd"):
            key = "%s:mlp/%s" % ("bwd" if "bwd" in s else "fwd", "mlp_bwd_kernel" if "bwd" in s else "mlp_fwd_kernel")
        else:
            key = "other/" + s.split("<")[0]
        cnt[s.split("<")[0]] = c + 1
        out.append(key)
    return out


def main():
    fetch, write, steps, outp = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    assert len(fetch) == len(write) and [r[1] for r in fetch] == [r[1] for r in write], "the two passes ran different launches"
    # a step starts with the launch that packs the weight operands of every layer (fgc_conv_pack: one per step)
    names = [r[1] for r in fetch]
    starts = [i for i, k in enumerate(names) if "pack_many_kernel" in k]
    assert len(starts) == steps, (len(starts), steps)
    bounds = starts + [len(names)]
    per_key, whole = {}, []
    for s in range(steps):
        a, b = bounds[s], bounds[s + 1]
        keys = label(names[a:b])
        tot_f = tot_w = 0.0
        for i, key in zip(range(a, b), keys):
            f, w = fetch[i][3] * 1024.0, write[i][3] * 1024.0
            tot_f += f
            tot_w += w
            if s == steps - 1 and not key.startswith("other/"):
                per_key[key] = {"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                                "grid": fetch[i][2]}
        whole.append({"FETCH_SIZE_bytes": tot_f, "WRITE_SIZE_bytes": tot_w, "hbm_bytes_per_step": 2 * tot_f + tot_w,
                      "launches": b - a})
    out = dict(per_key)
    out["whole_step"] = whole[-1]
    out["whole_step_all"] = whole
    out["method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes of tools/pmc_steps.py %d "
                     "(torus 250x200, 100000 facets, forward+backward+Adam); values of the last step; hbm = 2 x FETCH_SIZE "
                     "(gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; counter values are KB" % steps)
    # what the pass is valid for: bench.py quotes these figures only while the kernel sources hash to the same value
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha16
    out["meta"] = {"commit": sys.argv[5] if len(sys.argv) > 5 else None, "command": sys.argv[6] if len(sys.argv) > 6 else None,
                   "csrc_sha16": csrc_sha16()}
    json.dump(out, open(outp, "w"), indent=1)
    ws = whole[-1]
    print("whole step: FETCH %.1f MB x2 + WRITE %.1f MB = %.1f MB over %d launches" % (
        ws["FETCH_SIZE_bytes"] / 1e6, ws["WRITE_SIZE_bytes"] / 1e6, ws["hbm_bytes_per_step"] / 1e6, ws["launches"]))
    for k in sorted(per_key, key=lambda k: -per_key[k]["hbm_bytes_per_launch"])[:16]:
        v = per_key[k]
        print("%-52s fetch %7.1f MB  write %7.1f MB  hbm %7.1f MB" % (k, v["FETCH_SIZE_bytes"] / 1e6, v["WRITE_SIZE_bytes"] / 1e6,
                                                                    v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
