#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $OLDPWD/tools/valu_issue_bench.hip -o /tmp/valu_issue_bench && /tmp/valu_issue_bench ) > $O/valu_issue_bench.txt 2>&1
cat $O/valu_issue_bench.txt
for bs in "4096,4" "8192,4" "8192,6" "16384,12" "32768,12"; do
  echo "== RAYNET_HIP_BOX_SPLIT=$bs" >> $O/proxy_box_split.txt
  RAYNET_HIP_BOX_SPLIT=$bs NO_PROF=1 WORLDS=8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world" | cut -c1-50 >> $O/proxy_box_split.txt
  RAYNET_HIP_BOX_SPLIT=$bs CONFIG=config4 NO_PROF=1 WORLDS=8 timeout 600 python tools/shard_proxy.py 2>&1 | grep -E "^world" | cut -c1-50 | sed 's/^/c4 /' >> $O/proxy_box_split.txt
done
cat $O/proxy_box_split.txt
