"""Round 5: does generation 3 with the LDS-landed residual epilogue now beat generation 2 on the level-0 one-tile-column residual GEMMs
(N = 320, K <= 1280: the dispatcher's 'short rule' sends them to generation 2)?  Run twice: EW_G3_SHORT=0 / EW_G3_SHORT=1."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from exp42_epi_phase import lib, timeit, dense   # noqa: E402
for name, mk in [("dense 460800x320x320 res", lambda: dense(460800, 320, 320)), ("dense 460800x320x320 res+rb", lambda: dense(460800, 320, 320, True)),
                 ("dense 460800x320x1280 res", lambda: dense(460800, 320, 1280))]:
    fn = mk()
    t = min(timeit(fn, iters=6, warm=3) for _ in range(4))
    print(f"EW_G3_SHORT={os.environ.get('EW_G3_SHORT', '0')} {name:30s} {lib.ew_gemm_last_kernel().decode():24s} {t:7.1f} us", flush=True)
    del fn
    torch.cuda.empty_cache()
