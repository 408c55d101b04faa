#!/usr/bin/env bash
# PMC passes over one bench step (counters in their own runs, kernel-trace only).
# Usage on the GPU box: bash tools/pmc_passes.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pmc off --no-config4 --no-reference-shapes ${BENCH_ARGS:-}"
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_EA0_RDREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum"
do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : $SET"
done
