export TMPDIR=/tmp; cd /tmp
for set in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
rm -rf /tmp/pa; rocprofv3 --pmc $set --output-format csv -d /tmp/pa -o p -- python $GRAFT_REPO_ROOT/bench.py --pmc off --no-cpu-baseline --no-config4 --no-reference-shapes --steps 2 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pa/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sweep_map" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print({k: sum(v)/len(v) for k,v in agg.items()})
PY
done
