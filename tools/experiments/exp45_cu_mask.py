"""Round 5: the two rows of the CFG batch as two concurrent B = 1 forwards on CU-MASKED streams (128 CUs each, the
persistent kernels' CU budget halved), phase-shifted so that one row's HBM-bound passes can overlap the other's MFMA-bound GEMMs --
against the B = 2 forward on the whole chip, same box, alternating.  python tools/experiments/exp45_cu_mask.py"""
import copy, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evoworld_amd import _lib
from evoworld_amd.unet import UNetSpatioTemporalConditionModel

lib = _lib.load()
dev = "cuda"
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device=dev)
T, h, w = 25, 72, 128
g = torch.Generator(device=dev).manual_seed(0)
def inputs(B):
    x = torch.randn(B * T * h * w, 64, device=dev, dtype=torch.float16, generator=g)
    x[:, 18:] = 0
    return x, torch.randn(B, 1, 1024, device=dev, dtype=torch.float16, generator=g), torch.tensor([[6.0, 127.0, 0.02]] * B, device=dev)
x2, e2, a2 = inputs(2)
rows = [(x2[: T * h * w].contiguous(), 1.234, e2[:1].contiguous(), a2[:1].contiguous()), (x2[T * h * w:].contiguous(), 1.234, e2[1:].contiguous(), a2[1:].contiguous())]

def run_b2(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = unet.forward_nhwc(x2, 1.234, e2, a2, 2, T, h, w)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

def make_streams(layout):
    # layout "interleaved": bits [0,128) / [128,256) of the CU mask; "halves" is the same call -- which physical CUs a bit range names is the
    # driver's business (KFD spreads consecutive mask bits over the XCDs), both give two disjoint halves of the chip
    sa, sb = lib.ew_stream_create_cu_mask(0, 128), lib.ew_stream_create_cu_mask(128, 128)
    if not sa or not sb:
        raise SystemExit("ew_stream_create_cu_mask failed: " + lib.ew_last_error().decode())
    return torch.cuda.ExternalStream(sa), torch.cuda.ExternalStream(sb), (sa, sb)

units = [unet, copy.copy(unet)]
units[1]._gn_pool = None                      # per-forward scratch must not be shared by two forwards in flight (weights are)

def run_pair(n, sa, sb, offset_ms, budget):
    """n x (row 0 on stream sa || row 1 on stream sb); row 1 starts offset_ms later (a sleep kernel on its stream)"""
    lib.ew_set_cu_budget(budget)
    outs = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for r, s in ((0, sa), (1, sb)):
            with torch.cuda.stream(s):
                if r == 1 and offset_ms > 0:
                    torch.cuda._sleep(int(offset_ms * 2.0e6))          # ~2 GHz cycles
                outs[r] = units[r].forward_nhwc(*rows[r], 1, T, h, w)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    lib.ew_set_cu_budget(256)
    return dt, outs

for _ in range(2):
    run_b2(1)
ref_ms, ref = run_b2(3)
print(f"B=2 forward, whole chip: {ref_ms:.1f} ms", flush=True)
# B = 1 forwards back to back on the default stream (the CFG-pair code path on one GPU)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    o0 = units[0].forward_nhwc(*rows[0], 1, T, h, w); o1 = units[1].forward_nhwc(*rows[1], 1, T, h, w)
torch.cuda.synchronize()
print(f"two B=1 forwards in sequence, whole chip: {(time.perf_counter() - t0) / 2 * 1e3:.1f} ms", flush=True)
sa, sb, raw = make_streams("interleaved")
ds = torch.cuda.current_stream()
for budget in (128,):        # (never 256 on a 128-CU stream: a stream-K finisher would wait for a contributor that is not resident)
    for off in (0, 10, 25, 50):
        run_pair(1, sa, sb, off, budget)
        ms, outs = run_pair(3, sa, sb, off, budget)
        e = float((torch.cat([outs[0], outs[1]]).float() - ref.float()).norm() / ref.float().norm())
        print(f"two B=1 forwards on CU-masked streams (128 + 128 CUs), CU budget {budget}, row 1 offset {off:3d} ms: {ms:.1f} ms per pair  (vs B=2 output: rel-L2 {e:.1e})", flush=True)
# unmasked streams, for reference (round 3's exp19: serialises)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
run_pair(1, s1, s2, 0, 256)
ms, _ = run_pair(3, s1, s2, 0, 256)
print(f"two B=1 forwards on two UNMASKED streams, CU budget 256: {ms:.1f} ms per pair", flush=True)
b2, _ = run_b2(3)
print(f"B=2 forward, whole chip (again): {b2:.1f} ms", flush=True)
from evoworld_amd import ops
ops.streamk_check()
