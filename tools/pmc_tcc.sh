#!/usr/bin/env bash
# L2 hit / miss counters of the bench step's kernels for the current environment (e.g. RAYNET_RAY_TILE=64x4)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_tcc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pass1 -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pass1.log 2>&1
python $R/tools/pmc_summary.py $OUT | grep -A5 "== k_sweep_map\|== k_bp"
