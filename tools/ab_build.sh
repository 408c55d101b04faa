#!/usr/bin/env bash
# A/B helper for the GPU box: rebuild the library with extra -D flags and run the bench.
# usage: tools/ab_build.sh "<label>" "<extra hipcc flags>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/raynet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -fPIC -shared $2 raynet_hip.hip -o libraynet_hip.so 2>&1 | grep -E "error" | head -3
cd $R
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
