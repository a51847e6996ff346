#!/usr/bin/env python3
"""Benchmark of the facet-graph-convolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of the reference's loop body (train.py:558-575,619) on one
synthetic mesh: random rotation of inputs and ground truth, full 3-level graph U-Net + MLP forward,
normalisation, angular loss on 4000 sampled rows, full backward, TF1-Adam update.  fp32 throughout.

Workload (BASELINE.json configs[1]): torus 250 x 200 quads = 100 000 facets,
preprocessed natively (adjacency + 4 pairing levels + binary-tree order) before the timed region;
everything is resident in HBM when the clock starts.

N > 1: one process per GPU (torchrun).  Default (--mode shard): ONE mesh of N x 100 000 facets (torus
250N x 200) is facet-sharded over the N GPUs (shard.py): each rank owns a contiguous range of the coarsest
graph level and everything under it, halo rows are exchanged by RCCL all-to-all before every conv (and
s = dy/deg rows plus cross-edge d-logits in backward), the flat fp32 gradient is all-reduced once per
step.  Per-GPU work is fixed as N grows ("weak" scaling); --scaling strong shards the 100k mesh instead.
--mode replicas: each rank trains on its own 100k-facet mesh, gradient all-reduce only.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel's achieved TFLOP/s from hipEvent timings taken live in this process
  cpu_baseline  the oracle (reference-shaped torch CPU restatement) timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see facet_graph_convolution_amd/__init__.py

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0


def build_mesh(nu, nv, seed):
    from facet_graph_convolution_amd.meshgen import torus, add_noise
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    V, F = torus(nu, nv)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F, 0.2, seed=1 + seed), F, V, seed=seed)
    return ds, F.shape[0]


def kernel_flops(kind, n, nnz, cin, cout, M=9):
    """Algorithmic FLOPs of one launch: the dense contraction 2*n*M*cin*cout plus the per-edge
    aggregation 2*nnz*M*C (C = gathered width)."""
    gemm = 2.0 * n * M * cin * cout
    if kind == "fwd":          # gathers cin-wide rows
        return gemm + 2.0 * nnz * M * cin
    if kind == "bwd_logits":   # dz GEMM + per-edge dots over cin
        return gemm + 2.0 * nnz * M * cin
    if kind == "bwd_data":     # gathers cout-wide rows of s
        return gemm + 2.0 * nnz * M * cout
    if kind == "bwd_weight":   # r^T x
        return gemm
    raise ValueError(kind)


def algorithmic_bytes_fwd_bwd(net):
    """SURVEY.md §8d convention: each tensor crossing a layer boundary written once, read once per consumer;
    CSR read once per conv; weights and fused ops free.  Backward moves the same tensors as gradients plus
    the saved activations again (x2.45 of forward in the survey's accounting: 3879/1584)."""
    dims = {name: (n, nnz, cin, cout) for name, n, nnz, cin, cout in net.layer_dims()}
    n0, n1, n2 = dims["conv1"][0], dims["conv2"][0], dims["conv3"][0]
    d0, d1, d2 = (dims[k][1] / dims[k][0] for k in ("conv1", "conv2", "conv3"))
    fwd = 4 * (n0 * (6 + 32 + 1 + d0) + n1 * (32 + 64 + 1 + d1) + n2 * (64 + 128 + 1 + d2) +
               n2 * (128 + 128 + 1 + d2) + n1 * (32 + 64 + 1 + d1) + n1 * (128 + 64 + 1 + d1) +
               n0 * (16 + 32 + 1 + d0) + n0 * (64 + 32 + 1 + d0) + n0 * (32 + 3) + n1 * 32 + n2 * 64)
    return fwd, fwd * 3879.0 / 1584.0


def cpu_baseline(sample_faces=(140, 140)):
    """The oracle timed on this host: one forward+backward of the reference-shaped torch-CPU restatement on a
    39 200-facet torus (bounded: about 10 s of CPU work, so that the default bench run stays within minutes), up to 32
    host threads."""
    import torch
    from oracle import model_ref as R
    ds, F = build_mesh(sample_faces[0], sample_faces[1], seed=7)
    x = torch.tensor(ds.in_list[0].astype(np.float32))
    gt = torch.tensor(ds.gt_list[0].astype(np.float32))
    adjs = [torch.tensor(a.astype(np.int32)) for a in ds.adj_list[0]]
    threads = min(os.cpu_count() or 1, 32)   # more threads than this only add contention on these op sizes
    torch.set_num_threads(threads)
    params = [p.requires_grad_(True) for p in R.init_params(0)]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = torch.eye(3)
    t0 = time.time()
    loss, _ = R.train_loss(x, adjs, gt, params, samp, Rm)
    loss.backward()
    dt = time.time() - t0
    return {"value": F / dt, "unit": "facets/s", "cores": threads, "kind": "port",
            "sample": "oracle/model_ref.py (reference-shaped K-padded torch CPU fp32), 1 forward+backward of the "
                      "full net on a torus %dx%d = %d facets (N0=%d), %.1f s" % (sample_faces[0], sample_faces[1], F, x.shape[1], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nu", type=int, default=250)
    ap.add_argument("--nv", type=int, default=200)
    ap.add_argument("--graph", type=int, default=0,
                    help="replay the forward+backward enqueue as one hipGraph (1 %% faster than eager here: the step is "
                         "GPU-bound).  Needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which the package sets at import: with "
                         "the runtime's pre-built graph packets a replay after a stream synchronise computes garbage on "
                         "this ROCm stack (DESIGN.md section 6), so the timed default stays on eager launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-kernels", type=str, default="", help="write the full per-kernel table to this file")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard", help="multi-GPU decomposition")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." %
                         (args.gpus, args.gpus))
    # FGC_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (host-staged exchange)
    backend = os.environ.get("FGC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local_dev = local_rank % max(ndev, 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.utils import rand_rotation_matrix

    shard = world > 1 and args.mode == "shard"
    if shard:
        from facet_graph_convolution_amd.shard import ShardPlan, DistComm, graphs_to_host_csr
        nu = args.nu * world if args.scaling == "weak" else args.nu
        ds, F_total = build_mesh(nu, args.nv, seed=0)          # every rank builds the same mesh (seeded)
        plan = ShardPlan(graphs_to_host_csr(ds.adj_list[0]), rank, world)
        net = FacetDenoiser(dev, seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0], plan=plan,
                                                   comm=DistComm())
        n0 = ds.in_list[0].shape[1]            # samples are drawn over the WHOLE mesh, same stream on every rank
        rs = np.random.RandomState(100)
        F = F_total / world                    # facets per GPU (for the per-GPU accounting below)
        halo_frac = [net._mesh["nh"][l] / max(net._mesh["ns"][l], 1) for l in range(3)]
    else:
        ds, F = build_mesh(args.nu, args.nv, seed=rank)
        F_total = F * world
        net = FacetDenoiser(dev, seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
        n0 = ds.in_list[0].shape[1]
        rs = np.random.RandomState(100 + rank)
        halo_frac = None

    # per-step random inputs (train.py:561-565) for the whole run, uploaded once: inside the loop they are refreshed by
    # device-to-device copies (stream-ordered with the hipGraph replay)
    nsteps_total = args.warmup + args.steps
    samp_host = [rs.randint(n0, size=4000) for _ in range(nsteps_total)]
    rot_host = [rand_rotation_matrix(randnums=rs.uniform(size=3)) for _ in range(nsteps_total)]
    if not shard:
        S_all = torch.from_numpy(np.stack(samp_host).astype(np.int32)).to(dev)
        R_all = torch.from_numpy(np.stack(rot_host).astype(np.float32).reshape(nsteps_total, 9)).to(dev)
    torch.cuda.synchronize()
    counter = [0]

    def step():
        k = counter[0] % nsteps_total
        counter[0] += 1
        if shard:
            net.set_samples(samp_host[k])
            net.set_rotation(rot_host[k])
        else:
            net.set_step_inputs_device(S_all[k], R_all[k])
        net.forward_backward(rotate=True, capture=bool(args.graph) and not shard)
        if world > 1 and not shard:
            if backend == "nccl":
                dist.all_reduce(net.params.grad, op=dist.ReduceOp.AVG)
            else:
                g = net.params.grad.cpu()
                dist.all_reduce(g)
                net.params.grad.copy_(g / world)
        net.adam_step()
        if os.environ.get("FGC_BENCH_TRACE"):
            print("trace loss %.3f" % net.buffers["loss"][0].item(), file=sys.stderr)

    def sync_barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = net.buffers["loss"][0].item()

    # the same steps as hipGraph replays (one graph per step holds the whole forward+backward enqueue), untimed extra:
    # a second network from the same seed walks the same inputs, so its final loss must equal the eager one bit for bit
    hipgraph = None
    if world == 1 and not shard and not args.graph:
        net_g = FacetDenoiser(dev, seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])

        def step_g(k):
            net_g.set_step_inputs_device(S_all[k % nsteps_total], R_all[k % nsteps_total])
            net_g.forward_backward(rotate=True, capture=True)
            net_g.adam_step()

        for k in range(args.warmup):
            step_g(k)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for k in range(args.warmup, nsteps_total):
            step_g(k)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg
        loss_g = net_g.buffers["loss"][0].item()
        hipgraph = {"ms_per_step": tg / args.steps * 1e3, "loss_deg": loss_g, "matches_eager": loss_g == loss}
        del net_g

    # forward-only rate (BASELINE config 2 wording), untimed extra
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(max(3, args.steps // 2)):
        net.forward(rotate=False)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t1) / max(3, args.steps // 2) * 1e3
    if world > 1:
        dist.barrier()

    roofline = None
    kernels = {}
    if not args.no_roofline and not shard:
        # per-kernel durations from hipEvents recorded by the library around every launch (same stream), over
        # `steps` eager steps of the same work as the timed region
        net.profile_start()
        for k in range(args.steps):
            net.set_step_inputs_device(S_all[k % nsteps_total], R_all[k % nsteps_total])
            net.forward_backward(rotate=True, capture=False)
            net.adam_step()
        prof = net.profile_stop()
        dims = {name: (n, nnz, cin, cout) for name, n, nnz, cin, cout in net.layer_dims()}
        total_ms = sum(ms for _, ms in prof.values())
        rows = []
        for key, (cnt, ms) in prof.items():
            tag, kern = key.split("/", 1)
            phase, layer = (tag.split(":") + [""])[:2]
            kind = None
            if layer in dims:
                if "conv_fwd_kernel" in kern or "conv_w8_kernel<fwd>" in kern:
                    kind = "fwd"
                elif "conv_bwd_logits" in kern:
                    kind = "bwd_logits"
                elif "conv_bwd_data_kernel" in kern or "conv_w8_kernel<data>" in kern:
                    kind = "bwd_data"
                elif "gemm_tn_kernel" in kern and cnt == args.steps:
                    kind = "bwd_weight"
            avg_us = ms / cnt * 1e3
            fl = kernel_flops(kind, *dims[layer]) if kind else None
            if layer == "mlp" and kern in ("mlp_fwd_kernel", "mlp_bwd_kernel"):
                # 32 -> 1024 -> 3 per padded node; the backward recomputes the hidden layer and adds dW and dx
                fl = 2.0 * dims["conv1"][0] * 1024 * (32 + 3) * (3 if kern == "mlp_bwd_kernel" else 1)
            rows.append((ms, key, cnt, avg_us, fl))
        rows.sort(reverse=True)
        if args.dump_kernels and rank == 0:
            with open(args.dump_kernels, "w") as fh:
                for ms, key, cnt, avg_us, fl in rows:
                    fh.write("%-60s launches/step %4.1f  avg %9.2f us  per-step %9.2f us  share %5.2f%%  %s\n" % (
                        key, cnt / args.steps, avg_us, ms / args.steps * 1e3, 100 * ms / total_ms,
                        ("%.1f TFLOP/s" % (fl / (avg_us * 1e-6) / 1e12)) if fl else ""))
        for ms, key, cnt, avg_us, fl in rows[:12]:
            kernels[key] = {"launches": cnt, "avg_us": round(avg_us, 2), "share": round(ms / total_ms, 4),
                            "tflops": round(fl / (avg_us * 1e-6) / 1e12, 2) if fl else None}
        # the dominant kernel = the kernel function with the largest summed time per step (the d-logits kernel: seven
        # launches), reported through its longest launch
        fam = {}
        for r in rows:
            if r[4]:
                fam[r[1].split("/", 1)[1]] = fam.get(r[1].split("/", 1)[1], 0.0) + r[0]
        top = max(fam, key=fam.get) if fam else None
        # ... the launch with the most algorithmic work (ties: first by name, so that the choice does not flip between
        # two equal layers from run to run)
        cands = sorted((r for r in rows if r[4] and r[1].split("/", 1)[1] == top), key=lambda r: (-r[4], r[1]))
        dom = cands[0] if cands else None
        if dom:
            ms, key, cnt, avg_us, fl = dom
            ach = fl / (avg_us * 1e-6) / 1e12
            # HBM bytes per launch of that kernel from the PMC passes kept under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE,
            # gfx950 correction of MI355X_MICROARCH.md); null when the dominant kernel has no recorded pass
            traffic = None
            tpath = os.path.join(REPO, "profiles", "r1_traffic_dominant_kernel.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("kernel") == key and args.nu == 250 and args.nv == 200:
                    traffic = tj["hbm_bytes_per_launch"]
            roofline = {"bound": "mfma", "kernel": key, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "avg_kernel_us": round(avg_us, 2), "launch_flops": fl,
                        "eager_step_ms_sum_of_kernels": round(total_ms / args.steps, 3)}

    fwd_b, fb_b = algorithmic_bytes_fwd_bwd(net)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "facets/sec (fwd+bwd) on 100k-facet mesh",
            "value": F_total * args.steps / dt,
            "unit": "facets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong" if (shard and args.scaling == "strong") else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "torus %dx%d quads = %d facets per GPU (N0=%d padded nodes%s), full graph U-Net + MLP: "
                                   "rotate + forward + angular loss + backward + Adam, fp32%s" %
                                   (args.nu, args.nv, int(F), n0, " in the whole mesh" if shard else "",
                                    ", hipGraph replay" if (args.graph and not shard) else ""),
                       "parallelism": ("single GPU" if world == 1 else
                                       ("one %d-facet mesh facet-sharded over %d GPUs, halo all-to-all per conv + flat-gradient "
                                        "all-reduce (halo/owned rows per level on rank 0: %s)" %
                                        (F_total, world, ", ".join("%.3f" % h for h in halo_frac))) if shard else
                                       "1 mesh per GPU, flat-gradient all-reduce")},
            "loss_deg": loss,
            "hipgraph_replay": hipgraph,
            "forward_only_ms": fwd_ms,
            "forward_only_facets_per_s": F_total / (fwd_ms * 1e-3),
            "hbm_roofline_frac_whole_step": fb_b / (ms_step * 1e-3) / (PEAK_HBM_GBS * 1e9),
            "algorithmic_bytes_fwd_bwd": fb_b,
            "roofline": roofline,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
