"""GPU parity: libfgc graph convolution vs the golden fixtures (reference source) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONV_CASES = ["c1_raw", "c1_coarsened", "rand_5_7", "rand_32_64", "rand_128_64", "rand_nomask"]


def _params(cin, cout, seed, device):
    from oracle import model_ref as R
    return [p.to(device) for p in R.conv_params(cin, cout, seed)]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_matches_golden(golden_dir, case):
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    z = np.load(os.path.join(golden_dir, "conv_%s.npz" % case))
    z64 = np.load(os.path.join(golden_dir, "conv_%s_f64.npz" % case))
    dev = torch.device("cuda:0")
    x = torch.tensor(z["x"][0], device=dev)
    g = FacetGraph(z["adj"], dev)
    params = _params(x.shape[1], int(z["cout"]), int(z["seed"]), dev)
    y, _, _ = ops.conv_fwd(g, x, None, 0, params, bias_mask=(case != "rand_nomask"))
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    ref32, ref64 = z["y"][0], z64["y"][0]
    err_ref = np.abs(ref32 - ref64).max()
    err_gpu = np.abs(got - ref64).max()
    print("%s: |gpu-f64| %.3e  |ref32-f64| %.3e  |gpu-ref32| %.3e" % (case, err_gpu, err_ref, np.abs(got - ref32).max()))
    # fp32 tolerance: both fp32 evaluations must sit within 2e-6 of the float64 result
    assert err_gpu < 2e-6
