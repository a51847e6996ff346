"""MI355X-native facet graph convolution (see README.md)."""
import os

# hipGraph replays (FacetDenoiser.forward_backward(capture=True)) go wrong on ROCm 7.2 after a hipStreamSynchronize /
# hipDeviceSynchronize when the runtime has pre-built the graph's AQL packets (its default): the second replay that is
# enqueued behind another one after such a sync runs with stale state (tools/graph_inputs_probe.py: loss 88 deg instead
# of 33).  hipEventSynchronize does not trigger it, re-instantiating the graph cures it, and so does turning the packet
# pre-building off - which has to happen before the HIP runtime initialises, hence here.  Eager launches are unaffected.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
