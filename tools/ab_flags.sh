#!/usr/bin/env bash
# usage on the GPU box: tools/ab_flags.sh "<label>" "<extra hipcc flags>" [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
label=$1; flags=$2; shift 2
cd $R/raynet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -fPIC -shared $flags raynet_hip.hip -o libraynet_hip.so 2>&1 | grep -E "error" | head -3
cd $R
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --pmc table "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
