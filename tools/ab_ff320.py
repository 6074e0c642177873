"""A/B timing of the fused level-0 feed-forward (spatial and temporal epilogue forms) -- run with EW_LIB_PATH for another build."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from evoworld_amd import ops
g = torch.Generator().manual_seed(0)
rows, C, HID = 460800, 320, 1280
x = (torch.rand(rows, C, generator=g) * 2 - 1).half().cuda()
pack = ops.ff_pack((torch.randn(2 * HID, C, generator=g) * 0.05).cuda(), (torch.randn(2 * HID, generator=g) * 0.1).cuda(), (torch.randn(C, HID, generator=g) * 0.03).cuda())
b2 = (torch.randn(C, generator=g) * 0.1).half().cuda()
r1 = ops.Res.from_float((torch.rand(rows, C, generator=g) * 2 - 1).cuda())
r2 = ops.Res.from_float((torch.rand(rows, C, generator=g) * 2 - 1).cuda())
out = ops.Res.empty(rows, C, "cuda", True)
outp = torch.empty(rows, C, dtype=torch.float16, device="cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rep in range(2):
    print(f"ff320 spatial (r1 split, out split): {t(lambda: ops.ff_geglu320(x, pack, b2, out, r1=r1)):.1f} us   temporal blend (r1, r2 split, out fp16): {t(lambda: ops.ff_geglu320(x, pack, b2, outp, c_acc=0.5, r1=r1, c_r1=0.5, r2=r2, c_r2=0.5)):.1f} us", flush=True)
print("checksum", float(out.hi.float().sum()), float(outp.float().sum()))
