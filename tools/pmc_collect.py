"""Developer helper: HBM traffic per kernel launch and per whole step from two rocprofv3 PMC passes of tools/pmc_steps.py
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, both with --kernel-trace --output-format csv; counters never share
a pass with the tracing domains gpurun refuses).

    python tools/pmc_collect.py <fetch counter_collection.csv> <write counter_collection.csv> <steps> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section): hbm = 2 * FETCH + WRITE.  Launches are matched to layers by the fixed schedule of a
step (see `label`)."""
import csv
import json
import sys

FWD = ["conv1", "conv2", "conv3", "dconv3", "upconv2", "dconv2", "upconv1", "dconv1"]
BWD = ["dconv1", "upconv1", "dconv2", "upconv2", "dconv3", "conv3", "conv2", "conv1"]


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


def is_fwd_conv(s):
    # (conv_bfm_kernel: the bf16 kernels with the aggregation on the matrix pipe; same role, same profile key)
    return s.startswith("conv_w8_kernel<false") or s.startswith("conv_bfm_kernel<false")


def label(names):
    """tag/kernel key (as bench.py prints them) for every dispatch of ONE step.  The schedule is fixed: a forward layer
    starts with its logit-table launch (proj_mfma_kernel, or pair_transform_kernel for a layer in the pair form), a backward
    layer with its d-logits launch (conv_bwd_logits_*, pair_bwd_logits_kernel, conv_narrow_bwd_kernel); the backward pass
    begins at the MLP's backward kernel."""
    out = []
    fi = bi = -1
    bwd = False
    started = False      # a table launch opened the current forward layer
    for k in names:
        s = short(k)
        base = s.split("<")[0]
        key = "other/" + base
        if base.startswith("mlp_bwd") or base.startswith("loss_step"):
            bwd = True
        if not bwd:
            if base in ("proj_mfma_kernel", "proj_narrow_kernel", "pair_transform_kernel", "pair_transform_bf16_kernel"):
                fi += 1
                started = True
            elif is_fwd_conv(s) or base in ("pair_fwd_kernel", "conv_narrow_fwd_mma_kernel", "conv_narrow_fwd_kernel"):
                if not started:      # (the first layer's table can come with the step's housekeeping launch)
                    fi += 1
                started = False
            lay = FWD[min(max(fi, 0), 7)]
            if is_fwd_conv(s):
                key = "fwd:%s/conv_w8_kernel<fwd>" % lay
            elif base.startswith("pair_transform"):
                key = "fwd:%s/pair_transform_kernel" % lay
            elif base == "pair_fwd_kernel":
                key = "fwd:%s/pair_fwd_kernel" % lay
            elif base.startswith("mlp_fwd"):
                key = "fwd:mlp/mlp_fwd_kernel"
        else:
            if base.startswith("conv_bwd_logits") or base == "pair_bwd_logits_kernel" or base.startswith("conv_narrow_bwd"):
                bi += 1
            lay = BWD[min(max(bi, 0), 7)]
            if s.startswith("conv_w8_kernel<true") or s.startswith("conv_bfm_kernel<true"):
                key = "bwd:%s/conv_w8_kernel<data>" % lay
            elif base.startswith("conv_bwd_logits"):
                key = "bwd:%s/%s" % (lay, base)
            elif base == "pair_bwd_logits_kernel":
                key = "bwd:%s/pair_bwd_logits_kernel" % lay
            elif base.startswith("gemm_tn") and "group" in base:
                key = "bwd:reduce/gemm_tn_kernel:dW"       # (several per step: the last one's counters are kept)
            elif base.startswith("gemm_tn"):
                key = "bwd:%s/gemm_tn_kernel:dW" % lay
            elif base.startswith("mlp_bwd_dx"):
                key = "bwd:mlp/mlp_bwd_kernel<dx>"
            elif base.startswith("mlp_bwd_w"):
                key = "bwd:mlp/mlp_bwd_kernel<w>"
            elif base.startswith("mlp_bwd"):
                key = "bwd:mlp/mlp_bwd_kernel"
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
