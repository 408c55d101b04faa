#!/usr/bin/env bash
# A/B of variant builds of the library (build_variants/lib_<NAME>.so) against the product with
# tools/scatter_probe.py (kernel families by event pairs + wall step), each twice in turns.
# usage: VARIANTS="A B" CONFIGS="config2 config4" bash tools/r06_variants_probe.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
REC=$OUT/${1:-r06_x}_variants_probe.txt
: > $REC
for cfg in ${CONFIGS:-config2}; do
  for rep in 1 2; do
    for v in PRODUCT ${VARIANTS}; do
      if [ $v = PRODUCT ]; then unset RAYNET_HIP_LIB; else export RAYNET_HIP_LIB=$R/build_variants/lib_$v.so; fi
      echo "== $cfg $v (rep $rep)" >> $REC
      timeout 300 python tools/scatter_probe.py --config $cfg --levels auto 2>&1 >/dev/null | grep '^{' >> $REC
    done
  done
done
python3 - $REC <<'PY'
import sys, json
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith('=='): print(l, end=' ')
    elif l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['kernel_ms_per_step'])
PY
