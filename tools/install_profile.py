#!/usr/bin/env python3
"""Copy a profile_round.sh result set gpurun_out/<tag>_* into profiles/ and (config 2 only)
regenerate profiles/pmc_traffic.json from its PMC summary.
usage: tools/install_profile.py <tag> [<tag>_config4 ...]"""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in sys.argv[1:]:
    for f in ("bench.json", "kernel_stats_rocprofv3.txt", "pmc_counters.txt", "rocprof_bench_line.txt"):
        shutil.copy(os.path.join(R, "gpurun_out", "%s_%s" % (tag, f)), os.path.join(R, "profiles", "%s_%s" % (tag, f)))
    if "config4" in tag:
        continue
    txt = open(os.path.join(R, "profiles", tag + "_pmc_counters.txt")).read()
    names = {"k_bp": "bp", "k_depth": "depth", "k_scatter_box": "scatter", "k_sweep_map": "sweep_map",
             "k_traverse": "traverse"}
    out, raw = {}, {}
    for blk in txt.split("== ")[1:]:
        k = blk.split("\n")[0].strip()
        if k not in names:
            continue
        f = float(re.search(r"FETCH_SIZE\s+total\s+\d+\s+per launch\s+(\d+)", blk).group(1))
        w = float(re.search(r"WRITE_SIZE\s+total\s+\d+\s+per launch\s+(\d+)", blk).group(1))
        out[names[k]] = int((2 * f + w) * 1024)
        raw[names[k]] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w}
    out["_config"], out["_profile"] = "config2", tag
    out["_how"] = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc "
                   "passes (tools/pmc_passes.sh via tools/profile_round.sh, summary profiles/%s_pmc_counters.txt); "
                   "FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950; on k_bp the "
                   "corrected read side (2.56 GB) matches the algorithmic row reads (12 B x 211 M voxel visits = "
                   "2.53 GB), which calibrates it for this access pattern.  Launch sizes: sweep_map / traverse / bp "
                   "/ scatter cover the whole 5-image scene, depth one image" % tag)
    out["_raw"] = raw
    json.dump(out, open(os.path.join(R, "profiles", "pmc_traffic.json"), "w"), indent=1)
    d = json.load(open(os.path.join(R, "profiles", tag + "_bench.json")))
    print(tag, d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["frac"],
          d["roofline"]["avg_launch_ms"], d["path_roofline"],
          {k: v["total_ms_per_step"] for k, v in d["kernels"].items()})
