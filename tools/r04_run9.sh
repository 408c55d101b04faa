#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
bash tools/gloo_scaling.sh r04i > /dev/null 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04i_gloo_ranks.jsonl"):
    try:
        d=json.loads(l)
        print(d["n_gpus"], d["ms_per_step"], d["step_capture"], json.dumps(d["ranks"])[:900])
    except Exception as e: print("bad line", e, l[:200])
PY
tail -3 gpurun_out/r04i_gloo_8.err
timeout 600 python -m pytest tests/test_forward_pass_gpu.py -q -m gpu -k "ranks_without or exchange_pieces" 2>&1 | tail -3
