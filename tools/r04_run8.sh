#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
for cfg in config2 config4; do
CONFIG=$cfg STEPS=10 ROUNDS=3 PROF=1 timeout 900 python tools/ab_options.py base > $O/ab_${cfg}_plain.txt 2>&1
RAYNET_HIP_STEP_LISTS=1 CONFIG=$cfg STEPS=10 ROUNDS=3 PROF=1 timeout 900 python tools/ab_options.py base > $O/ab_${cfg}_steps.txt 2>&1
CONFIG=$cfg STEPS=10 ROUNDS=3 PROF=1 timeout 900 python tools/ab_options.py base > $O/ab_${cfg}_plain2.txt 2>&1
RAYNET_HIP_STEP_LISTS=1 CONFIG=$cfg STEPS=10 ROUNDS=3 PROF=1 timeout 900 python tools/ab_options.py base > $O/ab_${cfg}_steps2.txt 2>&1
for f in plain steps plain2 steps2; do echo "$cfg $f: $(tail -1 $O/ab_${cfg}_$f.txt)"; done
done
