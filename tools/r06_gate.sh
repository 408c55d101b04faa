#!/usr/bin/env bash
# round-6 gate of the footprint-staged plane sweep: times + L2 counters of the
# gathers alone, product order (0), plane-blocked tile (3) and footprint-staged tiles (20, 21, 22)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
for CFG in config2 config4; do
  timeout 900 python tools/sweep_gather_bench.py --config $CFG --variants 0,3,20,21,22,23 --out gpurun_out/r06_gate_$CFG.json 2>&1 | grep -v "amdgpu.ids" | tail -12
  for V in 0 3 20 21; do
    rm -rf /tmp/pmc_$V
    (cd /tmp && timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_$V -o p -- python $GRAFT_REPO_ROOT/tools/sweep_gather_bench.py --config $CFG --variants $V --once > /dev/null 2>&1)
    python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$V/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float)
for p in f:
    for r in csv.DictReader(open(p)):
        if "sgb" in r["Kernel_Name"] or "k_ray" in r["Kernel_Name"] or "k_tile" in r["Kernel_Name"]:
            tot[(r["Kernel_Name"][:60], r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print("$CFG variant $V", k[0], k[1], "%.1f M" % (v / 1e6))
PY
  done
done 2>&1 | tee gpurun_out/r06_gate_counters.txt
