"""Developer helper (GPU box, one GPU): what ONE collective call costs the compute stream on this platform, apart from the wire -
backend 'nccl' (= RCCL) at world size 1, the call path a multi-GPU run takes.  A loop of [small kernel on the compute stream;
one collective] is timed on the GPU's own clock (events on the compute stream) for: no collective at all; all_to_all_single /
all_reduce with async_op=False (torch >= 2.7 runs a synchronous op on the CURRENT stream: no cross-stream dependency); the same
with async_op=True followed at once by work.wait() (the op on RCCL's own stream: the compute stream's event is waited for there,
RCCL's event here - what shard.DistComm did for every exchange until round 6).  The difference between the last two is the
price of the two cross-stream dependencies; the difference to the empty loop is the fixed cost of a collective that has nobody
to talk to.  usage: python tools/rccl_call_cost_probe.py [iterations]"""
import os, sys, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
send = torch.ones(16384, device=dev)
recv = torch.empty_like(send)
small = torch.zeros(4096, device=dev)
red = torch.ones(4, device=dev)
cases = {
    "no collective": lambda: None,
    "all_to_all_single, async_op=False": lambda: dist.all_to_all_single(recv, send, [16384], [16384]),
    "all_to_all_single, async_op=True + wait()": lambda: dist.all_to_all_single(recv, send, [16384], [16384], async_op=True).wait(),
    "all_reduce (4 floats), async_op=False": lambda: dist.all_reduce(red),
    "all_reduce (4 floats), async_op=True + wait()": lambda: dist.all_reduce(red, async_op=True).wait(),
}
out = {}
for rep in range(3):
    for name, fn in cases.items():
        for _ in range(20):
            small.add_(1.0); fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(n_it):
            small.add_(1.0)
            fn()
        b.record()
        torch.cuda.synchronize()
        host = (time.perf_counter() - t0) / n_it * 1e6
        out[name] = min(out.get(name, (1e9, 0))[0], a.elapsed_time(b) / n_it * 1e3), host
base = out["no collective"][0]
print("torch %s, %d iterations of [one small kernel; one collective], minimum of 3 passes; world size 1" % (torch.__version__, n_it))
for name, (us, host) in out.items():
    print("  %-48s %7.2f us per iteration on the compute stream (%+6.2f over the empty loop), host %.1f us" % (name, us, us - base, host))
dist.destroy_process_group()
