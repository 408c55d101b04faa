#!/usr/bin/env python3
"""Copy a profile_round.sh result set gpurun_out/<tag>_* into profiles/ and regenerate this
configuration's entry of profiles/pmc_traffic.json (per kernel family and launch: HBM-side
bytes, VALU instructions, wavefronts) from its PMC summary.  Tags that contain "config4"
describe BASELINE.json configs[3], every other tag configs[1].
usage: tools/install_profile.py <tag> [<tag>_config4 ...]"""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"k_bp": "bp", "k_depth": "depth", "k_scatter_box": "scatter", "k_sweep_map": "sweep_map",
         "k_traverse": "traverse"}
path = os.path.join(R, "profiles", "pmc_traffic.json")
try:
    table = json.load(open(path))
    if "config2" not in table and "config4" not in table:
        table = {}
except Exception:
    table = {}
table["_how"] = (
    "per kernel family and launch, from separate rocprofv3 --pmc passes of bench.py (tools/pmc_passes.sh via "
    "tools/profile_round.sh; summaries profiles/<profile>_pmc_counters.txt): traffic_bytes = (2*FETCH_SIZE + "
    "WRITE_SIZE) KiB -- FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950; on k_bp "
    "the corrected read side matches the algorithmic row reads (12 B x voxel visits), which calibrates it for "
    "this access pattern; valu_insts = SQ_INSTS_VALU (a wave64 VALU instruction occupies its SIMD for 4 "
    "cycles: issue floor = valu_insts * 4 / (1024 SIMDs * 2.4 GHz)); waves = SQ_WAVES")


def per_launch(blk, counter):
    m = re.search(counter + r"\s+total\s+\d+\s+per launch\s+(\d+)", blk)
    return float(m.group(1)) if m else None


for tag in sys.argv[1:]:
    for f in ("bench.json", "kernel_stats_rocprofv3.txt", "pmc_counters.txt", "rocprof_bench_line.txt"):
        shutil.copy(os.path.join(R, "gpurun_out", "%s_%s" % (tag, f)), os.path.join(R, "profiles", "%s_%s" % (tag, f)))
    config = "config4" if "config4" in tag else "config2"
    txt = open(os.path.join(R, "profiles", tag + "_pmc_counters.txt")).read()
    kernels = {}
    for blk in txt.split("== ")[1:]:
        k = blk.split("\n")[0].strip()
        if k not in NAMES:
            continue
        f, w = per_launch(blk, "FETCH_SIZE"), per_launch(blk, "WRITE_SIZE")
        kernels[NAMES[k]] = dict(traffic_bytes=int((2 * f + w) * 1024), fetch_size_kb=f, write_size_kb=w,
                                 valu_insts=per_launch(blk, "SQ_INSTS_VALU"), waves=per_launch(blk, "SQ_WAVES"))
    table[config] = dict(profile=tag, kernels=kernels)
    d = json.load(open(os.path.join(R, "profiles", tag + "_bench.json")))
    print(tag, d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["frac"],
          d["roofline"]["avg_launch_ms"], d["path_roofline"],
          {k: v["total_ms_per_step"] for k, v in d["kernels"].items()})
json.dump(table, open(path, "w"), indent=1)
