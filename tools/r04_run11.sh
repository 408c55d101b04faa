#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_forward_pass_gpu.py tests/test_saturated_golden.py -x -q -m gpu -k "plan_path or captured or deterministic or saturated or sharded" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for t in off 2048 4096 8192 16384; do
  if [ $t = off ]; then export RAYNET_SCATTER_ITEMS=0; else export RAYNET_SCATTER_ITEMS=1; export RAYNET_SCATTER_TARGET=$t; fi
  echo "== items $t" >> $O/proxy_items.txt
  WORLDS=1,8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world" | cut -c1-140 >> $O/proxy_items.txt
  NO_PROF=1 WORLDS=1,2,4,8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world|ceiling" | cut -c1-60 >> $O/proxy_items.txt
done
cat $O/proxy_items.txt
