#!/usr/bin/env bash
# One-stop profile of the current state on the GPU box: bench line (with CPU baseline),
# rocprofv3 kernel-trace statistics of the same command, PMC passes.  Only small text
# summaries are left under gpurun_out/ (raw traces are deleted).
# usage: bash tools/profile_round.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
# extra bench.py arguments (e.g. "--config config4") for every leg
BARGS=${BENCH_ARGS:-}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python bench.py --steps 10 --warmup 2 $BARGS 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc table $BARGS > $OUT/${TAG}_rocprof_bench_line.txt 2>/tmp/prof_kt.log
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py $DB $OUT/${TAG}_kernel_stats_rocprofv3.txt > /dev/null; fi
CSV=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
if [ -n "$CSV" ]; then head -40 $CSV > $OUT/${TAG}_kernel_stats.csv; fi
rm -rf /tmp/prof_kt
bash $R/tools/pmc_passes.sh /tmp/pmc > $OUT/${TAG}_pmc_passes.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc $OUT/${TAG}_pmc_counters.txt > /dev/null
rm -rf /tmp/pmc
ls -la $OUT
