#!/bin/bash
mkdir -p gpurun_out/r02d
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_inference.py tests/test_gpu_reprojection.py -q -m gpu -x > gpurun_out/r02d/pytest.log 2>&1
tail -5 gpurun_out/r02d/pytest.log
# CPU baseline: bounded sample vs one complete full-size forward (extrapolation error)
python - > gpurun_out/r02d/cpu_baseline.log 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
a = bench.cpu_baseline(frames=4)
print(json.dumps(a))
b = bench.cpu_baseline(frames=50)
print(json.dumps(b))
print("extrapolation error (sample/full frames/s - 1):", a["value"] / b["value"] - 1)
PY
cat gpurun_out/r02d/cpu_baseline.log
