"""Round 6: does capturing the U-Net forward (about 1150 launches) into a hipGraph buy anything?  Eager launches through ctypes on one stream against
a replay of the captured graph, same process, alternating; outputs compared bit for bit (the stream-K hand-over clears its flags for replays)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evoworld_amd.unet import UNetSpatioTemporalConditionModel
from evoworld_amd import ops
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
B, T, h, w = 2, 25, 72, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16, generator=g)
x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16, generator=g)
ids = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")
ts = torch.tensor([1.234], device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        ref = unet.forward_nhwc(x, ts, ehs, ids, B, T, h, w)
    torch.cuda.synchronize()
    ref = ref.clone()
    graph = torch.cuda.CUDAGraph()
    t0 = time.time()
    with torch.cuda.graph(graph, stream=s):
        out = unet.forward_nhwc(x, ts, ehs, ids, B, T, h, w)
    torch.cuda.synchronize()
    print(f"capture + instantiate: {time.time() - t0:.2f} s")

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best
    for rnd in range(3):
        e = timed(lambda: unet.forward_nhwc(x, ts, ehs, ids, B, T, h, w))
        r = timed(lambda: graph.replay())
        print(f"round {rnd}: eager {e:.2f} ms   graph replay {r:.2f} ms")
    graph.replay(); torch.cuda.synchronize()
    print("graph output == eager output:", bool(torch.equal(out, ref)))
    ops.streamk_check()
