set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --pmc off --no-config4 --no-reference-shapes > $OUT/r06_i_tl_bench_line.txt 2>/tmp/tl.log
DB=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/step_timeline.py $DB $OUT/r06_i_step_timeline_1gpu.txt > /dev/null
tail -3 /tmp/tl.log
cat $OUT/r06_i_step_timeline_1gpu.txt
