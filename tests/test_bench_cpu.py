"""bench.py host logic that needs no GPU: workload presets, the N > 1 self-launch (a fresh torchrun child, never an
exec), and that a failing child is reported as a failure instead of a hang or a silent success."""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_config_presets_follow_the_survey():
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.nu, a.nv, a.multi_scale, a.scaling, a.what, a.dtype) == (1, 250, 200, False, "weak", "train", "f32")
    a = bench.parse_args(["--config", "c3"])
    assert (a.nu, a.nv, a.what, a.dtype) == (250, 100, "train", "bf16")          # 50 000 facets, bf16 storage
    assert bench.parse_args(["--config", "c3", "--dtype", "f32"]).dtype == "f32"
    a = bench.parse_args(["--config", "c4", "--gpus", "8"])
    assert (a.nu * a.nv * 2, a.scaling, a.what) == (1000000, "strong", "train")   # ONE 1M-facet mesh over the ranks
    a = bench.parse_args(["--config", "c5", "--gpus", "8"])
    assert (a.nu * a.nv * 2, a.multi_scale, a.scaling, a.what) == (500000, True, "strong", "denoise")
    a = bench.parse_args(["--nu", "40", "--nv", "30", "--multi-scale"])
    assert (a.nu, a.nv, a.what) == (40, 30, "denoise")


def test_launcher_command_is_the_drivers_torchrun_line():
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--nu", "48", "--steps", "7", "--multi-scale"], 29999)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == os.path.join(REPO, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_family_names_group_by_kernel_function():
    import bench
    f = bench.family_of
    assert f("conv_w8_kernel<fwd>") == f("conv_w8_kernel<data>") == "conv_w8"
    assert f("conv_bwd_logits_deep_kernel") == f("conv_bwd_logits_mfma_kernel") == "conv_bwd_logits"
    assert f("mlp_bwd_kernel") == "mlp" and f("gemm_tn_stream_kernel") == "gemm_tn"


def test_self_launch_without_a_gpu_fails_loudly():
    """No GPU here: both ranks of the child die in FacetDenoiser ("no CPU fallback").  The parent must come back with a
    non-zero exit code and no JSON line - not hang, not print a number."""
    env = dict(os.environ, FGC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--nu", "12", "--nv", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=REPO)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "one more group" not in r.stderr          # eager launches are the default of N > 1: nothing to retry without


def test_self_launch_retries_once_with_eager_launches_when_a_graph_run_dies_without_a_line():
    """--graph 1 (hipGraph segments between the exchanges): a rank group that ends without a result line is followed by ONE
    fresh child group with --graph 0.  Here both groups die (no GPU): the parent says what it did and fails."""
    env = dict(os.environ, FGC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--nu", "12", "--nv", "10", "--no-cpu-baseline", "--graph", "1"], env=env, capture_output=True,
                       text=True, timeout=600, cwd=REPO)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.stderr.count("one more group with --graph 0") == 1


def test_algorithmic_bytes_match_the_survey_formula():
    """SURVEY.md section 8d per-node forward bytes (1 584 B per padded level-0 node at d = 10.72/8.17/7.92)."""
    import bench

    class Net:
        def layer_dims(self):
            n0, n1, n2 = 123472, 30868, 7717
            z0, z1, z2 = 1323472, 252236, 61155
            return [("conv1", n0, z0, 6, 32), ("conv2", n1, z1, 32, 64), ("conv3", n2, z2, 64, 128),
                    ("dconv3", n2, z2, 128, 128), ("upconv2", n1, z1, 128, 64), ("dconv2", n1, z1, 128, 64),
                    ("upconv1", n0, z0, 64, 32), ("dconv1", n0, z0, 64, 32)]
    fwd, fb = bench.algorithmic_bytes_fwd_bwd(Net())
    assert abs(fwd / 123472 - 1584) < 12 and abs(fb / fwd - 3879 / 1584) < 1e-9
    fwd16, _ = bench.algorithmic_bytes_fwd_bwd(Net(), elem=2)
    assert 0.5 * fwd < fwd16 < 0.62 * fwd      # activations halve, the CSR and the 6 / 3-channel ends do not
