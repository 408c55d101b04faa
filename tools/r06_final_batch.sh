#!/usr/bin/env bash
# the round's final-state records: tools/r06_final_batch.sh <tag>   (through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_e}
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -6) > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
timeout 900 python tools/fullsize_parity.py > /dev/null 2>&1; cp gpurun_out/fullsize_parity.json gpurun_out/${TAG}_fullsize_parity.json
python - <<PY
import json; d=json.load(open('gpurun_out/${TAG}_fullsize_parity.json')); print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='differing_pixels'}) for k,v in d.items()})
PY
R=$PWD; cd /tmp; rm -rf /tmp/kstats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $R/bench.py --pmc off --no-cpu-baseline --no-config4 --no-reference-shapes > /dev/null 2>&1; cd $R
cp /tmp/kstats/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_rocprofv3.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats_rocprofv3.csv")))
for r in rows[:7]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
timeout 1500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
NO_PROF=1 WORLDS=1,2,4,8 ALL_RANKS=1 timeout 900 python tools/shard_proxy.py > gpurun_out/${TAG}_shard_proxy_config2.txt 2>&1; grep -i "ceiling\|slowest" gpurun_out/${TAG}_shard_proxy_config2.txt | tail -6
