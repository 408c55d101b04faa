cd $GRAFT_REPO_ROOT
for sb in 0 1 0 1; do
  RAYNET_SLAB_BOXES=$sb python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('slab_boxes=$sb', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
done
for sb in 0 1; do
  RAYNET_SLAB_BOXES=$sb python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config4 slab_boxes=$sb', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5
