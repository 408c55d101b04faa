#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
for K in 1 2 3 5; do
  echo "== exchange_pieces=$K" >> $O/proxy_pieces.txt
  RAYNET_EXCHANGE_PIECES=$K NO_PROF=1 WORLDS=1,8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world" | cut -c1-50 >> $O/proxy_pieces.txt
  RAYNET_EXCHANGE_PIECES=$K CONFIG=config4 NO_PROF=1 WORLDS=8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world" | cut -c1-50 | sed 's/^/c4 /' >> $O/proxy_pieces.txt
done
cat $O/proxy_pieces.txt
