# A/B of two prebuilt libraries: tools/libraynet_hip_base.so vs the in-tree one
cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_new.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"; }
for i in 1 2 3; do
cp tools/libraynet_hip_base.so raynet_amd/csrc/libraynet_hip.so; run base
cp /tmp/lib_new.so raynet_amd/csrc/libraynet_hip.so; run new
done
cp tools/libraynet_hip_base.so raynet_amd/csrc/libraynet_hip.so; run "config4 base" --config config4
cp /tmp/lib_new.so raynet_amd/csrc/libraynet_hip.so; run "config4 new" --config config4
