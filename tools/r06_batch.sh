#!/usr/bin/env bash
# round-6 GPU batch: tools/r06_batch.sh <tag> [tests|bench|all]   (run through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_a}
WHAT=${2:-all}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$WHAT" = tests ] || [ "$WHAT" = all ]; then
  (timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -25) > gpurun_out/${TAG}_pytest_gpu.log
  cat gpurun_out/${TAG}_pytest_gpu.log | tail -8
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  timeout 900 python bench.py --pmc off --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  tail -c 3000 gpurun_out/${TAG}_bench.json
  tail -5 gpurun_out/${TAG}_bench.err
fi
