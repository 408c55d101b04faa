#!/usr/bin/env bash
# round-4 profile of the current state: tools/r04_profile.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_a}
export TMPDIR=/tmp
bash tools/profile_round.sh $TAG > /dev/null 2>&1
BENCH_ARGS="--config config4" bash tools/profile_round.sh ${TAG}_config4 > /dev/null 2>&1
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.log
NO_PROF=1 WORLDS=1,2,4,8 ALL_RANKS=1 timeout 900 python tools/shard_proxy.py > gpurun_out/${TAG}_shard_proxy_config2.txt 2>&1
CONFIG=config4 NO_PROF=1 WORLDS=1,2,4,8 ALL_RANKS=1 timeout 1200 python tools/shard_proxy.py > gpurun_out/${TAG}_shard_proxy_config4.txt 2>&1
WORLDS=1,8 timeout 600 python tools/shard_proxy.py > gpurun_out/${TAG}_shard_proxy_config2_kernels.txt 2>&1
CONFIG=config4 WORLDS=1,8 timeout 600 python tools/shard_proxy.py > gpurun_out/${TAG}_shard_proxy_config4_kernels.txt 2>&1
grep ceiling gpurun_out/${TAG}_shard_proxy_config*.txt
timeout 900 python tools/fullsize_parity.py > gpurun_out/${TAG}_fullsize_parity.log 2>&1
tail -4 gpurun_out/${TAG}_fullsize_parity.log
python - <<PY
import json
for t in ("$TAG", "${TAG}_config4"):
    try:
        d = json.loads(open("gpurun_out/%s_bench.json" % t).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(t, d["ms_per_step"], d["value"], r["bound"], r["frac"], r["avg_launch_ms"], r.get("valu_frac"), r.get("valu_cycles_per_inst"), d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
        print("   ", {k: (v["total_ms_per_step"], v["algorithmic_GBps"]) for k, v in d["kernels"].items()})
    except Exception as e:
        print(t, "bench parse failed", e)
PY
