"""-m 'not gpu': bench.py's launch contract that can be checked without a device: `--gpus N` from a plain invocation must
refuse to run when fewer than N devices answer (it never silently falls back to one rank), a WORLD_SIZE that disagrees with
--gpus is an error, and the N>1 clip sharding + weight-broadcast helpers work over gloo with two ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_n_refuses_when_fewer_devices_answer():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 2 but only 0 device(s) answer" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                                     "MASTER_PORT": "29999", "EW_BENCH_SKIP_INIT": "1"})
    assert r.returncode != 0
    assert "--gpus 1 but WORLD_SIZE=2" in (r.stderr + r.stdout)
