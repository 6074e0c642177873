"""Round 3: s_memtime segment trace of the fused feed-forward kernel (library built with -DFF_TRACE, EW_LIB_PATH).  Stamps of block 0,
per chunk and wave: 0 phase-1 start, 1 up-projection issued, 2 after vmcnt(0), 3 after the barrier, 4 GEGLU done, 5 phase-2 end.
Prints the mean segment lengths (shader clocks) over the chunks of tiles 2..N for every wave."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib, ops  # noqa: E402

M, C = 460800, 320
g = torch.Generator().manual_seed(0)
w1 = ((torch.rand(2560, C, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
b1 = ((torch.rand(2560, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
w2 = ((torch.rand(C, 1280, generator=g) * 2 - 1) / 1280 ** 0.5).half().cuda()
b2 = ((torch.rand(C, generator=g) * 2 - 1) / 36).half().cuda()
x = torch.randn(M, C, generator=g).half().cuda()
h = ops.Res.from_float(torch.randn(M, C, generator=g).cuda())
pack = ops.ff_pack(w1, b1, w2)
out = ops.Res.empty(M, C, "cuda", True)
lib = _lib.load()
NW = int(os.environ.get("FF_NWV", "8"))
buf = torch.zeros(4096 * NW * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.ff_geglu320(x, pack, b2, out, r1=h)
torch.cuda.synchronize()
lib.ew_ff_set_trace.argtypes = [ctypes.c_void_p]
lib.ew_ff_set_trace(buf.data_ptr())
ops.ff_geglu320(x, pack, b2, out, r1=h)
torch.cuda.synchronize()
lib.ew_ff_set_trace(None)
t = buf.cpu().reshape(4096, NW, 8)
nch = int((t[:, 0, 0] != 0).sum())
print("chunks traced", nch)
t = t[:nch].double()
names = ["P1 issue (0->1)", "vmcnt wait (1->2)", "barrier (2->3)", "GEGLU (3->4)", "down+DMA (4->5)", "to next P1 (5->0')"]
sel = [c for c in range(80, nch - 1) if c % 40 not in (0, 39)]
for w in range(NW):
    seg = []
    for k in range(5):
        seg.append(float((t[sel, w, k + 1] - t[sel, w, k]).mean()))
    seg.append(float((t[[c + 1 for c in sel], w, 0] - t[sel, w, 5]).mean()))
    tot = float((t[[c + 1 for c in sel], w, 0] - t[sel, w, 0]).mean())
    print(f"wave {w}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, seg)) + f"   chunk {tot:6.0f}")
c0 = 120
print("absolute stamps of chunk", c0, "relative to wave 0 stamp 0:")
for w in range(NW):
    print(w, [int(v - t[c0, 0, 0]) for v in t[c0, w, :6]])
tile = float((t[160, 0, 0] - t[120, 0, 0]))
print("tile-to-tile (40 chunks + flush + epilogue):", tile, "clocks; per chunk", tile / 40)
