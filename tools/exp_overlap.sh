cd $GRAFT_REPO_ROOT
for ov in 0 1 0 1; do
  RAYNET_HIP_OVERLAP=$ov python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('overlap=$ov', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
done
RAYNET_HIP_OVERLAP=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config4 overlap=1', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
RAYNET_HIP_OVERLAP=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config4 overlap=0', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
