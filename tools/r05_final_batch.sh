export TMPDIR=/tmp
(python -m pytest tests -q -m gpu 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -4) > gpurun_out/r05_i_pytest_gpu.log; cat gpurun_out/r05_i_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/fullsize_parity.py > /dev/null 2>&1; cp gpurun_out/fullsize_parity.json gpurun_out/r05_fullsize_parity.json
python -c "
import json; d=json.load(open('gpurun_out/r05_fullsize_parity.json')); print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='differing_pixels'}) for k,v in d.items()})"
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $GRAFT_REPO_ROOT/bench.py --pmc off --no-cpu-baseline --no-config4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
cp /tmp/kstats/*kernel_stats.csv gpurun_out/r05_i_kernel_stats_rocprofv3.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r05_i_kernel_stats_rocprofv3.csv")))
for r in rows[:6]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
python bench.py > gpurun_out/r05_i_bench.json 2> gpurun_out/r05_i_bench.err; tail -c 600 gpurun_out/r05_i_bench.json
