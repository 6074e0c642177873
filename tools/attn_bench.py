"""Stand-alone timing of ew_attn_spatial_f16 at the level-0 shape (250 problems x S=9216 x 64) for counter runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evoworld_amd import ops
n_seq, S, heads = int(os.environ.get("NSEQ", 50)), int(os.environ.get("S", 9216)), 5
C, rows = heads * 64, n_seq * S
g = torch.Generator().manual_seed(0)
qk = torch.randn(rows, 2 * C, generator=g).half().cuda()
vt = torch.randn(C, rows, generator=g).half().cuda()
o = torch.empty(rows, C, dtype=torch.float16, device="cuda")
iters = int(os.environ.get("ITERS", 3))
qk2 = (qk.float() * ops.QK_LOG2_PRESCALE).half()          # what the projection epilogue hands ew_attn_spatial_log2_f16
o2 = torch.empty_like(o)
fl = 4.0 * n_seq * heads * S * S * 64
for rep in range(int(os.environ.get("REPS", 2))):
    for name, fn in (("attn_spatial     ", lambda: ops.attn_spatial(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)),
                     ("attn_spatial_log2", lambda: ops.attn_spatial_log2(qk2, qk2[:, C:], vt, o2, n_seq, S, heads, 2 * C, rows, C))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"{name} n_seq={n_seq} S={S}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF/s", flush=True)
d = (o.float() - o2.float()).norm() / o.float().norm()
print(f"log2 form vs scale-and-shift form: rel-L2 {d:.2e}")
