"""Round 5 (the alternative deliverable): cycle anatomy of the level-0 spatial attention (log2 form) from s_memtime stamps.
Library built with `make -C evoworld_amd/csrc attn_trace` (EW_LIB_PATH=.../libevoworld_hip_attn_trace.so).  One workgroup (block 4001: mid-launch,
all CUs busy) stamps, per 64-key tile and wave: 0 tile start, 1 the 8 QK^T MFMAs issued (K fragments read, next tile's global loads issued),
2 P ready (wait for the MFMA results + 32 v_exp + 16 cvt_pkrtz + 16 v_dot2), 3 the 8 P.V MFMAs issued, 4 next tile written to LDS (global-load
wait + 6 ds_write), 5 after the tile's barrier.  The script also times the launch: 18000 workgroups over 256 CUs x 3 resident = 23.4 workgroup
generations of 144 tiles each, which gives the tick rate of the counter (ticks per microsecond) next to the per-phase tick counts."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib, ops  # noqa: E402

n_seq, S, heads = 50, 9216, 5
C, rows = heads * 64, n_seq * S
g = torch.Generator().manual_seed(0)
qk = (torch.randn(rows, 2 * C, generator=g) * ops.QK_LOG2_PRESCALE).half().cuda()
vt = torch.randn(C, rows, generator=g).half().cuda()
o = torch.empty(rows, C, dtype=torch.float16, device="cuda")
lib = _lib.load()
run = lambda: ops.attn_spatial_log2(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device="cuda")
lib.ew_attn_set_trace.argtypes = [ctypes.c_void_p]
lib.ew_attn_set_trace(buf.data_ptr())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record()
torch.cuda.synchronize()
lib.ew_attn_set_trace(None)
ms = s.elapsed_time(e)
t = buf.cpu().reshape(256, 4, 8).double()
nt = int((t[:, 0, 0] != 0).sum())
print(f"launch {ms:.3f} ms ({4.0 * n_seq * heads * S * S * 64 / ms / 1e9:.0f} TF/s with the stamps in); tiles traced {nt} of {S // 64}")
sel = list(range(8, nt - 2))
names = ["QK^T issue + K reads (0->1)", "softmax: MFMA wait + exp/cvt/dot2 (1->2)", "P.V issue + V reads (2->3)", "global wait + ds_write (3->4)", "barrier (4->5)", "to next tile (5->0')"]
per_tile = float((t[sel[-1] + 1, 0, 0] - t[sel[0], 0, 0]) / (len(sel)))
us_per_tile = ms * 1e3 / (n_seq * heads * (S // 128) / (256 * 3)) / (S // 64)
print(f"s_memtime ticks per tile (wave 0): {per_tile:.1f}; launch-derived time per tile {us_per_tile * 1e3:.0f} ns -> {per_tile / us_per_tile:.0f} ticks per microsecond")
for w in range(4):
    seg = [float((t[sel, w, k + 1] - t[sel, w, k]).mean()) for k in range(5)]
    seg.append(float((t[[c + 1 for c in sel], w, 0] - t[sel, w, 5]).mean()))
    tot = sum(seg)
    print(f"wave {w}: " + "  ".join(f"{n} {v:6.1f} ({v / tot * 100:4.1f} %)" for n, v in zip(names, seg)) + f"   tile {tot:6.1f} ticks")
print("absolute stamps of tile 40 relative to wave 0's stamp 0:")
for w in range(4):
    print(w, [int(v - t[40, 0, 0]) for v in t[40, w, :6]])
