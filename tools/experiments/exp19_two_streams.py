"""CFG halves on two HIP streams: the uncond / cond halves of the batch are independent, so two host threads can each drive a
B=1 forward on its own stream; kernels of one half then fill the tile-quantisation tails and launch gaps of the other.
Compares one B=2 forward against two concurrent B=1 forwards (same total work)."""
import os, sys, threading, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd.unet import UNetSpatioTemporalConditionModel

torch.manual_seed(0)
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
B, T, h, w = 2, 25, 72, 128
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16); x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16)
added = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")
rows = T * h * w

def fwd_full():
    return unet.forward_nhwc(x, 1.234, ehs, added, B, T, h, w)

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]
def half(i):
    with torch.cuda.stream(streams[i]):
        outs[i] = unet.forward_nhwc(x[i * rows:(i + 1) * rows], 1.234, ehs[i:i + 1], added[i:i + 1], 1, T, h, w)

def fwd_split():
    ev = torch.cuda.Event(); ev.record()
    for s in streams: s.wait_event(ev)
    th = [threading.Thread(target=half, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    return torch.cat(outs, 0)

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

ref = fwd_full(); torch.cuda.synchronize()
got = fwd_split(); torch.cuda.synchronize()
print("rel-L2 split vs full:", float((got.float() - ref.float()).norm() / ref.float().norm()))
for rnd in range(2):
    print(f"one stream, B=2: {timeit(fwd_full):7.1f} ms    two streams, 2 x B=1: {timeit(fwd_split):7.1f} ms", flush=True)
# host-side cost of issuing one forward (no sync): is the CPU far enough ahead?
torch.cuda.synchronize(); t0 = time.perf_counter(); fwd_full(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host issue time of one B=2 forward: {(t1 - t0) * 1e3:.1f} ms")
