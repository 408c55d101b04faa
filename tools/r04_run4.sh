#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $OLDPWD/tools/valu_issue_bench.hip -o /tmp/valu_issue_bench && /tmp/valu_issue_bench ) > $O/valu_issue_bench.txt 2>&1
cat $O/valu_issue_bench.txt
timeout 1500 python -m pytest tests/test_forward_pass_gpu.py -x -q -m gpu -k "rccl or sharded" > $O/pytest_dist.log 2>&1
tail -3 $O/pytest_dist.log
NO_PROF=1 WORLDS=1,2,4,8 timeout 900 python tools/shard_proxy.py > $O/proxy_c2.txt 2>&1
grep -E "^world|ceiling" $O/proxy_c2.txt | cut -c1-60
NO_PROF=1 CONFIG=config4 WORLDS=1,2,4,8 timeout 900 python tools/shard_proxy.py > $O/proxy_c4.txt 2>&1
grep -E "^world|ceiling" $O/proxy_c4.txt | cut -c1-60
CONFIG=config4 WORLDS=1,8 timeout 900 python tools/shard_proxy.py > $O/proxy_c4_prof.txt 2>&1
grep -E "^world" $O/proxy_c4_prof.txt | cut -c1-170
# L2 misses of the config-4 sweep under three schedules
for t in "" 8x8 32x32; do
  OUT=/tmp/pmc_$t; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && RAYNET_SWEEP_TILE=$t timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pass1 -o p -- python $OLDPWD/bench.py --config config4 --steps 1 --warmup 1 --no-cpu-baseline --pmc off > $OUT/pass1.log 2>&1 )
  echo "sweep_tile=[$t]" >> $O/pmc_c4_sweep_tiles.txt
  python tools/pmc_summary.py $OUT | grep -A4 "== k_sweep_map" >> $O/pmc_c4_sweep_tiles.txt
done
cat $O/pmc_c4_sweep_tiles.txt
