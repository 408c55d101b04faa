#!/usr/bin/env bash
# usage: pmc_ab.sh <label> "<flags>"
R=${GRAFT_REPO_ROOT:-/root/repo}
label=$1; flags=$2
cd $R/raynet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -fPIC -shared $flags raynet_hip.hip -o libraynet_hip.so 2>&1 | grep -E "error" | head -3
OUT=$R/gpurun_out/pmc_$label
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES"
do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1
done
python $R/tools/pmc_summary.py $OUT | grep -A20 "== k_sweep_map" | grep -v "^== k_t"
find $OUT -name "*.csv" -size +1M -delete; rm -rf $OUT/pass*/*/*.db 2>/dev/null
