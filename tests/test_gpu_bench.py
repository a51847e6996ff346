"""bench.py on the GPU box: the contract line at N = 1, the N = 2 sharded path started from a plain shell (two ranks
on this one GPU over gloo: the same DistComm code a multi-GPU RCCL run uses, host-staged), the multi-scale mode."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SMALL = ["--steps", "2", "--warmup", "1", "--nu", "48", "--nv", "40", "--no-cpu-baseline", "--repeats", "2"]


def _bench(args, env=None, timeout=900):
    e = dict(os.environ, **(env or {}))
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout, cwd=REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    # extras that run AFTER the line was printed report on stderr (N > 1: the strong-scaling reading), and the record as it
    # stood before any extra ran is there too
    for l in r.stderr.splitlines():
        if l.startswith("bench: strong-scaling extra: "):
            j["_strong_after_line"] = json.loads(l[len("bench: strong-scaling extra: "):])
        if l.startswith("bench: headline record ") and ": {" in l:
            j["_early_record"] = json.loads(l[l.index(": {") + 2:])
    return j


def test_single_gpu_line_has_the_contract_fields_and_family_rooflines():
    j = _bench(SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert abs(j["value"] - 48 * 40 * 2 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    assert j["hipgraph_replay"]["matches_eager"] is True
    # a mesh this small is timed as ONE replayed hipGraph per step (its launches take the host longer than the GPU), and says so
    assert j["hipgraph_replay"]["timed_region"] == "hipgraph" and "hipGraph replay" in j["config"]["workload"]
    e = _bench(SMALL + ["--graph", "0"])
    assert e["hipgraph_replay"]["timed_region"] == "eager" and "hipGraph replay" not in e["config"]["workload"]
    fams = [f["family"] for f in j["families"]]
    assert len(fams) == 3 and "conv_w8" in fams
    # `roofline` names the launch that takes the most time per step, with the best launch of its family beside it
    rl = j["roofline"]
    assert rl["kernel"] in j["kernels"] and rl["selected_by"].startswith("largest time per step")
    assert max(k["avg_us"] * k["launches"] for k in j["kernels"].values() if k["tflops"]) == \
        j["kernels"][rl["kernel"]]["avg_us"] * j["kernels"][rl["kernel"]]["launches"]
    assert rl["family_best"]["frac"] >= rl["frac"] and 0 < rl["frac"] < 1 and len(j["repeats_ms_per_step"]) == 2
    # the record was on stderr before the bf16 extras and the CPU baseline ran, with the same measurement
    assert j["_early_record"]["value"] == j["value"] and j["_early_record"]["roofline"]["kernel"] == rl["kernel"]
    # PMC traffic is quoted only from a pass taken on THIS build's kernel sources and on the 250 x 200 workload
    assert j["roofline"]["traffic"] is None and j["roofline"]["traffic_source"]
    assert j["roofline"]["launches_per_step"] > 20 and j["startup_s"]["preprocess_s"] > 0 and j["world_check"] is None


def test_two_ranks_from_a_plain_shell_over_gloo():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts the ranks itself."""
    j = _bench(["--gpus", "2"] + SMALL, env={"FGC_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert "facet-sharded over 2 GPUs" in j["config"]["parallelism"] and "world size 2" in j["config"]["parallelism"]
    assert j["exchange"]["collectives_per_step"] > 0 and j["exchange"]["bytes_sent_per_step"] > 0
    assert 0 < j["loss_deg"] < 180
    # the N > 1 line is complete: rank 0's hipEvent roofline (a split layer counts with the sum of its launches), the
    # step's collectives one by one, start-up times, and the world as the back end's own all-reduce counts it
    assert j["roofline"] is not None and 0 < j["roofline"]["frac"] < 1 and len(j["families"]) == 3
    assert j["cpu_baseline"] is None                  # (the contract times the CPU on rank 0 at N = 1 only)
    per = j["exchange"]["per_collective"]
    assert len(per) == j["exchange"]["collectives_per_step"] == 17
    assert [c["kind"] for c in per].count("all_reduce") == 3 and all(c["ms"] > 0 for c in per)
    assert j["world_check"] == {"backend": "gloo", "ranks_in_all_reduce": 2, "get_world_size": 2}
    assert set(j["startup_s"]) == {"preprocess_s", "shard_plan_s", "bind_s"}
    # the default of N > 1 is eager launches (bench.py: graph_mode); no retry happened
    assert "eager launches" in j["config"]["parallelism"] and j["hipgraph_replay"] is None and j["retry_note"] is None
    # how much of the schedule overlaps its exchanges was measured on the job's own collectives before the warm-up (every rank
    # reads the same all-reduced times, so every rank chose alike); a pinned threshold or hipGraph segments: no tuning
    tune = j["split_tune"]
    assert set(tune["ms_per_step"]) == {"1024/window", "1024/behind", "none/behind", "256/window", "64/window"}
    assert all(v > 0 for v in tune["ms_per_step"].values()) and tune["default"] == "1024/window" and tune["passes"] == 2
    assert tune["chosen"] in tune["ms_per_step"] and "schedule tuning" in j["timeline"]
    if tune["chosen"] != tune["default"]:
        assert tune["ms_per_step"][tune["chosen"]] < 0.97 * tune["ms_per_step"][tune["default"]]
    # --graph 1: the same schedule replayed from hipGraphs, one per stretch of launches between two exchanges
    # (stretches WITH launches only: two requests back to back leave no graph; this small mesh splits no layer)
    e = _bench(["--gpus", "2", "--graph", "1"] + SMALL, env={"FGC_BENCH_BACKEND": "gloo"})
    assert e["hipgraph_replay"]["graphs_per_step"] >= 12 and e["hipgraph_replay"]["eager_ms_per_step"] > 0
    assert e["hipgraph_replay"]["timed_region"] == "hipgraph" and "hipGraph segments" in e["config"]["parallelism"]
    assert e["split_tune"] is None
    # weak scaling: the mesh has twice the facets of the single-GPU run
    assert "%d facets" % (2 * 48 * 40 * 2) in j["config"]["workload"]
    # ... and the same ranks report the OTHER reading of the metric beside it: the ONE single-GPU-sized mesh sharded over
    # them (strong scaling), timed with the same barriers after the weak region
    # (measured after the line left - a failing rank in there must not cost the weak-scaling value - and reported on stderr)
    assert "note" in j["strong"]
    st = j["_strong_after_line"]
    assert st["scaling"] == "strong" and st["facets"] == 48 * 40 * 2 and st["steps"] == 2 and 0 < st["loss_deg"] < 180
    assert abs(st["value"] - st["facets"] / (st["ms_per_step"] * 1e-3)) < 1e-6 * st["value"]
    assert j["also"] is None


def test_multi_scale_denoising_mode_sharded_and_single():
    a = _bench(["--multi-scale"] + SMALL)
    assert a["metric"].startswith("facets/sec (multi-scale") and a["loss_deg"] is None
    b = _bench(["--gpus", "2", "--multi-scale", "--scaling", "strong"] + SMALL, env={"FGC_BENCH_BACKEND": "gloo"})
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["exchange"]["collectives_per_step"] > 0


def test_two_ranks_bf16_storage_over_gloo():
    """BASELINE config 3 sharded: the bf16-storage network over two ranks; the halo rows travel as bf16, so the same
    exchanges move fewer bytes than the fp32 run of the same mesh."""
    a = _bench(["--gpus", "2"] + SMALL, env={"FGC_BENCH_BACKEND": "gloo"})
    b = _bench(["--gpus", "2", "--dtype", "bf16"] + SMALL, env={"FGC_BENCH_BACKEND": "gloo"})
    assert b["dtype"] == "bf16" and b["n_gpus"] == 2 and 0 < b["loss_deg"] < 180
    assert b["exchange"]["collectives_per_step"] == a["exchange"]["collectives_per_step"]
    # (on this small mesh the flat-gradient all-reduce, fp32 in both, is most of the bytes)
    assert b["exchange"]["bytes_sent_per_step"] < a["exchange"]["bytes_sent_per_step"]
