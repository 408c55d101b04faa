#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rm -rf /tmp/tl && NO_PROF=1 WORLDS=8 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o tl -- python $R/tools/shard_proxy.py > $R/$O/proxy_traced.txt 2>&1 )
DB=$(find /tmp/tl -name "*.db" | head -1)
python tools/step_timeline.py $DB $O/step_timeline_rank8_captured.txt > /dev/null
cat $O/step_timeline_rank8_captured.txt
