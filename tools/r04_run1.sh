#!/usr/bin/env bash
# round 4, first GPU session: the new host-side pieces
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_forward_pass_gpu.py -x -q -m gpu -k "captured or maps_of or sharded or rccl or plan_path or pixel_order or deterministic" > gpurun_out/r04a/pytest_new.log 2>&1
tail -5 gpurun_out/r04a/pytest_new.log
timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --pmc off > gpurun_out/r04a/bench_c2.json 2> gpurun_out/r04a/bench_c2.err
tail -c 600 gpurun_out/r04a/bench_c2.err; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04a/bench_c2.json").read().strip().splitlines()[-1])
    print("config2", d["ms_per_step"], d["value"], d["step_capture"], d["roofline"]["avg_launch_ms"], {k:v["total_ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e: print("bench parse failed", e)
PY
RAYNET_CAPTURE=0 timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --pmc off > gpurun_out/r04a/bench_c2_eager.json 2>/dev/null
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04a/bench_c2_eager.json").read().strip().splitlines()[-1])
    print("config2 eager", d["ms_per_step"], d["value"], d["step_capture"])
except Exception as e: print("bench parse failed", e)
PY
NO_PROF=1 WORLDS=1,8 ALL_RANKS=1 timeout 600 python tools/shard_proxy.py > gpurun_out/r04a/proxy_c2_captured.txt 2>&1
tail -14 gpurun_out/r04a/proxy_c2_captured.txt
WORLDS=1,8 timeout 600 python tools/shard_proxy.py > gpurun_out/r04a/proxy_c2_eager_prof.txt 2>&1
tail -6 gpurun_out/r04a/proxy_c2_eager_prof.txt
NO_PROF=1 RAYNET_GATHER=all WORLDS=1,8 timeout 600 python tools/shard_proxy.py > gpurun_out/r04a/proxy_c2_gather_all.txt 2>&1
tail -4 gpurun_out/r04a/proxy_c2_gather_all.txt
