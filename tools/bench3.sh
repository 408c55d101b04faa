# usage on the GPU box: bash tools/bench3.sh <label>   -- three bench.py runs, per-kernel ms per step
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'ms/step', d['ms_per_step'], ' '.join('%s=%.3f'%(k,v['total_ms_per_step']) for k,v in d['kernels'].items()))"; done
