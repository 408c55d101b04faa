#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
NO_PROF=1 WORLDS=1,8 ALL_RANKS=1 timeout 900 python tools/shard_proxy.py > $O/proxy_c2.txt 2>&1
grep -E "^world|ceiling" $O/proxy_c2.txt | cut -c1-52
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['ms_per_step'], d['value'])"
