R=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -fPIC -shared"
cd $R/raynet_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -I $R/include raynet_hip.hip -o /tmp/v_base.so &
/opt/rocm/bin/hipcc $FLAGS -DRN_SWEEP_SGPR_VIEWS=0 -I $R/include raynet_hip.hip -o /tmp/v_sg0.so &
wait
cd /tmp; export TMPDIR=/tmp
for v in base sg0; do
  rm -rf /tmp/pm_$v
  LIB=/tmp/v_$v.so CONFIG=config4 STEPS=1 ROUNDS=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm_$v -o p -- python $R/tools/ab_options.py base > /dev/null 2>&1
  python - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("/tmp/pm_%s/**/*counter_collection.csv" % v, recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sweep_map" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print(v, {k: "%.0fM/launch (%d)" % (tot[k] / n[k] / 1e6, n[k]) for k in sorted(tot)})
PY
done
