#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_forward_pass_gpu.py -x -q -m gpu -k "captured or maps_of or rccl" > gpurun_out/r04b/pytest_new.log 2>&1
tail -3 gpurun_out/r04b/pytest_new.log
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --pmc off > gpurun_out/r04b/bench_c2.json 2> gpurun_out/r04b/bench_c2.err
tail -c 300 gpurun_out/r04b/bench_c2.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04b/bench_c2.json").read().strip().splitlines()[-1])
    print("config2", d["ms_per_step"], d["value"], d["step_capture"], d["roofline"]["avg_launch_ms"])
except Exception as e: print("bench parse failed", e)
PY
CONFIG=config4 STEPS=5 ROUNDS=3 PROF=1 timeout 1200 python tools/ab_options.py base sweep_tile=16x16 sweep_tile=32x32 sweep_tile=32x16 sweep_tile=16x32 sweep_tile=8x8 sweep_tile=64x8 > gpurun_out/r04b/ab_c4_tiles.txt 2>&1
cat gpurun_out/r04b/ab_c4_tiles.txt | tail -8
CONFIG=config4 STEPS=5 ROUNDS=3 PROF=1 timeout 1200 python tools/ab_options.py base sweep_xcd_chunk=256 sweep_xcd_chunk=512 sweep_xcd_chunk=1024 sweep_xcd_chunk=4096 "sweep_tile=32x32,sweep_xcd_chunk=1024" "sweep_tile=32x32,sweep_xcd_chunk=4096" > gpurun_out/r04b/ab_c4_chunks.txt 2>&1
cat gpurun_out/r04b/ab_c4_chunks.txt | tail -8
for cap in off auto; do
NO_PROF=1 RAYNET_CAPTURE=$cap WORLDS=1,8 ALL_RANKS=1 timeout 600 python tools/shard_proxy.py > gpurun_out/r04b/proxy_c2_capture_$cap.txt 2>&1
grep -E "world 8|ceiling" gpurun_out/r04b/proxy_c2_capture_$cap.txt | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' | tr '\n' ';'; echo
done
