#!/usr/bin/env bash
# A/B of two prebuilt libraries on the GPU box: tools/libraynet_hip_base.so (built from the
# last commit before calling gpurun) against the in-tree one; usage: bash tools/ab_so.sh [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_new.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"; }
for i in $(seq ${1:-3}); do
cp tools/libraynet_hip_base.so raynet_amd/csrc/libraynet_hip.so; run base
cp /tmp/lib_new.so raynet_amd/csrc/libraynet_hip.so; run new
done
cp tools/libraynet_hip_base.so raynet_amd/csrc/libraynet_hip.so; run "config4 base" --config config4
cp /tmp/lib_new.so raynet_amd/csrc/libraynet_hip.so; run "config4 new" --config config4
