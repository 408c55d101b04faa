#!/usr/bin/env bash
# memory-pipeline counters (TCP / UTCL1) of the bench step's kernels
# (the TA_* set hangs rocprofv3 on this image until the timeout: left out)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_mem; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $SET"
done
python $R/tools/pmc_summary.py $OUT | grep -A22 "== k_sweep_map" | grep -v "== k_t"
