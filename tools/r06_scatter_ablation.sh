#!/usr/bin/env bash
# Round 6: where the box scatter's time goes NOW (timing-only variant builds of the library,
# build_variants/lib_<NAME>.so built with -DRN_ABL_* / -DRN_PREFETCH from
# tools/experiments/r06_scatter_ablation.patch), configs 2 and 4, tools/scatter_probe.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
REC=$OUT/r06_j_scatter_ablation.txt
: > $REC
for cfg in config2 config4; do
  for v in PRODUCT ${VARIANTS:-NO_FLUSH_ATOMIC NO_FLUSH NO_LDS NO_LDS_NO_FLUSH LOADS_ONLY PREFETCH}; do
    if [ $v = PRODUCT ]; then unset RAYNET_HIP_LIB; else export RAYNET_HIP_LIB=$R/build_variants/lib_$v.so; fi
    echo "== $cfg $v" >> $REC
    timeout 300 python tools/scatter_probe.py --config $cfg --levels auto 2>&1 >/dev/null | grep '^{' >> $REC
  done
done
cat $REC
