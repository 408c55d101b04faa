#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
CONFIG5=1 timeout 600 python tools/train_bench.py > $O/train_bench_config5.txt 2>&1
tail -3 $O/train_bench_config5.txt
timeout 1500 python -m pytest tests/test_config5_gpu.py tests/test_models.py tests/test_similarity_reference.py tests/test_exact_build_gpu.py tests/test_abi.py -x -q -m gpu > $O/pytest_misc.log 2>&1
tail -5 $O/pytest_misc.log
