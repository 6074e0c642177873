"""Round 5: de-phasing the persistent workgroups of generation 3 (debug hook, bits 4-5 = who waits, bits 8-15 = us): do the memory-bound
epilogues of one half of the chip overlap the MFMA-bound K loops of the other?  python tools/experiments/exp43_stagger.py
Result (profiles/r05_b_exp43_stagger.txt): no -- a forced offset costs its own length, the free one (workgroups with one tile fewer) is inside noise.
The kernel-side hook was removed again after the measurement (it is in commit 427ef50's successor only: `git log -S"de-phase the persistent"`)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from exp42_epi_phase import lib, timeit, dense, conv, convt   # noqa: E402

cases = [("dense 115200x640x640 rb", lambda: dense(115200, 640, 640, True), (8, 14, 20, 27)),
         ("dense 28800x1280x1280 rb", lambda: dense(28800, 1280, 1280, True), (10, 20, 30, 40)),
         ("convT 460800x320x960", lambda: convt(2, 25, 9216, 320), (5, 10, 20))]
# protocol: configurations interleaved (base first and between), several rounds, best of each -- the first measurement of a shape runs on a
# colder clock state than the later ones (first version of this script: a no-op configuration "won" 7-11 % against the base measured first)
for name, mk, uss in cases:
    fn = mk()
    cfgs = [0] + [(3 << 4) | (us << 8) for us in uss] + [(1 << 4) | (uss[1] << 8)]
    best = {c: 1e9 for c in cfgs}
    for rnd_ in range(4):
        for c in cfgs + [0]:
            lib.ew_set_gemm_debug(c)
            best[c] = min(best[c], timeit(fn, iters=6, warm=2))
    lib.ew_set_gemm_debug(0)
    print(f"{name:36s} base {best[0]:7.1f} us | fewer-tiles WGs wait: " + "  ".join(f"{(c >> 8)}us {best[c]:6.1f}" for c in cfgs[1:-1])
          + f" | odd CUs wait {uss[1]}us {best[cfgs[-1]]:6.1f}", flush=True)
    del fn
    torch.cuda.empty_cache()
