"""MI355X-native facet graph convolution (see README.md)."""
import os

# hipGraph replays (FacetDenoiser.forward_backward(capture=True)) go wrong on ROCm 7.2 after a hipStreamSynchronize /
# hipDeviceSynchronize when the runtime has pre-built the graph's AQL packets (its default): the second replay that is
# enqueued behind another one after such a sync runs with stale state (tools/graph_inputs_probe.py: loss 88 deg instead
# of 33).  hipEventSynchronize does not trigger it, re-instantiating the graph cures it, and so does turning the packet
# pre-building off - which has to happen before the HIP runtime initialises, hence here.  Eager launches are unaffected.
import sys as _sys

_VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_before = os.environ.get(_VAR)
os.environ.setdefault(_VAR, "0")
_torch = _sys.modules.get("torch")
_hip_was_up = bool(_torch is not None and _torch.cuda.is_initialized())
# False when the switch cannot have been in effect at HIP initialisation: the caller set another value, or touched
# torch.cuda before importing this package without having exported the variable itself.  forward_backward(capture=True)
# refuses to replay graphs then (it would compute garbage silently, see above).
GRAPH_REPLAY_SAFE = os.environ[_VAR] == "0" and (not _hip_was_up or _before == "0")


def require_graph_replay_safe():
    if not GRAPH_REPLAY_SAFE or os.environ.get(_VAR) != "0":
        raise RuntimeError(
            "hipGraph replay is unsafe in this process: %s was not '0' when the HIP runtime initialised (value at "
            "package import: %r, HIP already up: %s).  Export %s=0 before the first torch.cuda call, or import "
            "facet_graph_convolution_amd before touching torch.cuda; eager launches (capture=False) are unaffected."
            % (_VAR, _before, _hip_was_up, _VAR))
