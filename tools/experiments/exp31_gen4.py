"""Generation 4 (ping-pong main loop) against generation 3: bit-level agreement and speed, in one process."""
import ctypes, os, sys, math
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib, ops
import tools.bench_kernels as B
lib = _lib.load()
lib.ew_set_gemm_generation.argtypes = [ctypes.c_int]

def run_dense(M, N, K, act=0, rb=False):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    rbv = B.rnd(M // 7001 + 1, N) if rb else None
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=torch.float16, device="cuda")
    return lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, act=act, rowbias=rbv, ld_rowbias=N, rows_per_group=7001), out, 2.0 * M * N * K

def run_conv(Nimg, C, O, H, W, up=0, c2=0, rb=False):
    x = B.rnd(Nimg * H * W, C)
    x2 = B.rnd(Nimg * H * W, c2) if c2 else None
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    w, b = B.rnd(O, 9 * (C + c2)) * 0.02, B.rnd(O)
    M = Nimg * Ho * Wo
    rbv = B.rnd(Nimg, O) if rb else None
    out = torch.empty(M, O, dtype=torch.float16, device="cuda")
    return lambda: ops.gemm(x, w, out, M=M, N=O, c1=C, lda=C, a2=x2, c2=c2, lda2=c2, bias=b, mode=ops.A_CONV3X3, rowbias=rbv, ld_rowbias=O,
                            rows_per_group=Ho * Wo, conv=(Nimg, H, W, Ho, Wo, 1, up)), out, 2.0 * M * O * 9 * (C + c2)

cases = [("dense ragged 51237x320x320", run_dense(51237, 320, 320)), ("dense rb 25700x640x448", run_dense(25700, 640, 448, rb=True)),
         ("geglu L0", run_dense(460800, 2560, 320, act=2)), ("geglu L1", run_dense(115200, 5120, 640, act=2)), ("geglu L2", run_dense(28800, 10240, 1280, act=2)),
         ("qkv L0", run_dense(460800, 960, 320)), ("qkv L1", run_dense(115200, 1920, 640)), ("dense 28800x1280x5120", run_dense(28800, 1280, 5120)),
         ("conv L0 320+rb", run_conv(50, 320, 320, 72, 128, rb=True)), ("conv L0 cat640", run_conv(50, 320, 320, 72, 128, c2=320)),
         ("conv L1 640", run_conv(50, 640, 640, 36, 64)), ("conv L1 up", run_conv(50, 640, 640, 36, 64, up=1)),
         ("conv L2 1280+rb", run_conv(50, 1280, 1280, 18, 32, rb=True)), ("conv L2 cat2560", run_conv(50, 1280, 1280, 18, 32, c2=1280))]
for name, (fn, out, fl) in cases:
    lib.ew_set_gemm_generation(3); fn(); torch.cuda.synchronize(); ref = out.clone(); k3 = lib.ew_gemm_last_kernel().decode()
    lib.ew_set_gemm_generation(4); out.zero_(); fn(); torch.cuda.synchronize(); k4 = lib.ew_gemm_last_kernel().decode()
    same = torch.equal(out, ref)
    diff = float((out.float() - ref.float()).abs().max())
    t3 = t4 = 1e9
    for rnd in range(2):
        lib.ew_set_gemm_generation(3); t3 = min(t3, B.timeit(fn))
        lib.ew_set_gemm_generation(4); t4 = min(t4, B.timeit(fn))
    print(f"{name:28s} {k3:22s}->{k4:22s} identical={same} maxdiff={diff:.3g}  gen3 {t3:6.3f} ms {fl / t3 / 1e9:6.0f} TF/s   gen4 {t4:6.3f} ms {fl / t4 / 1e9:6.0f} TF/s  {100 * (t3 / t4 - 1):+5.1f}%", flush=True)
lib.ew_set_gemm_generation(3)
