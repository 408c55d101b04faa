#!/usr/bin/env python3
"""k_sweep_map's EXECUTED instruction budget per phase.

Builds of the library whose k_sweep_map wavefronts end after phase n (-DRN_SWEEP_STOP=n,
raynet_kernels.h) are run under `rocprofv3 --pmc SQ_INSTS_VALU ...`; the difference between two
consecutive builds is what ONE phase executes -- loops, the IEEE fall-backs and divergent tails
counted as they ran, not as they stand in the text.  The full build is also counted by instruction
class (SQ_INSTS_VALU_* counters), the dynamic mix tools/valu_mix.py could only estimate.

    python tools/sweep_phase_budget.py --build            # here: hipcc, ~1 min per variant
    gpurun -- python tools/sweep_phase_budget.py --run    # on the GPU box -> gpurun_out/r05_sweep_phase_budget.json
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
HERE = os.path.dirname(os.path.abspath(__file__))
STOPS = [(1, "segment / count / voxel-row loads"), (12, "projection into N views (exact index arithmetic)"),
         (2, "8 load rounds: offset exchange, gathers, pair products, 8-lane folds"), (3, "softmax"),
         (4, "planes -> voxels (prefix-max scan, plane table, interpolation)"), (45, "clip + renormalise"),
         (0, "first BP iteration folded in (occupancy, two scans, suffix scan, messages) + stores")]
CLASS_COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                  "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU",
                  "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"]


def lib_for(stop):
    return os.path.join(HERE, "libraynet_hip_stop%d.so" % stop)


def build():
    from raynet_amd import _lib

    def one(stop):
        _lib.build(force=True, extra_flags=["-DRN_SWEEP_STOP=%d" % stop], out=lib_for(stop))
        return stop
    with ThreadPoolExecutor(3) as ex:
        for s in ex.map(one, [s for s, _ in STOPS if s != 0]):
            print("built", lib_for(s), flush=True)


def counters(lib, names, config, tag):
    out = "/tmp/sweep_budget_%s" % tag
    env = dict(os.environ, TMPDIR="/tmp")
    if lib:
        env["RAYNET_HIP_LIB"] = lib
    res = {}
    for i in range(0, len(names), 4):            # a few counters per pass
        sub = names[i:i + 4]
        subprocess.run("rm -rf %s; cd /tmp && rocprofv3 --pmc %s --output-format csv -d %s -o p -- "
                       "python %s/bench.py --config %s --pmc off --no-cpu-baseline --no-config4 --steps 2 > /dev/null 2>&1"
                       % (out, " ".join(sub), out, REPO, config), shell=True, env=env, check=False)
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_sweep_map" in row["Kernel_Name"]:
                    d = res.setdefault(row["Counter_Name"], [])
                    d.append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in res.items()}, {k: len(v) for k, v in res.items()}


def run(config):
    import bench
    cfg = bench.CONFIGS[config]
    rays = cfg["views"] * cfg["H"] * cfg["W"]
    rep = {"config": config, "rays_per_launch": rays, "phases": [], "what": __doc__.split("\n\n")[1]}
    prev = 0.0
    for stop, what in STOPS:
        c, n = counters(lib_for(stop) if stop else None, ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES"],
                        config, "stop%d" % stop)
        valu = c.get("SQ_INSTS_VALU", 0.0)
        launches_per_step = 1 if config == "config2" else 2
        per_ray = valu * launches_per_step / rays
        rep["phases"].append({"stop_after": stop, "phase": what, "valu_per_ray_cumulative": round(per_ray, 1),
                              "valu_per_ray": round(per_ray - prev, 1),
                              "lds_per_ray_cumulative": round(c.get("SQ_INSTS_LDS", 0.0) * launches_per_step / rays, 1),
                              "salu_per_ray_cumulative": round(c.get("SQ_INSTS_SALU", 0.0) * launches_per_step / rays, 1),
                              "launches_seen": n.get("SQ_INSTS_VALU", 0)})
        prev = per_ray
        print(rep["phases"][-1], flush=True)
    c, _ = counters(None, CLASS_COUNTERS, config, "classes")
    rep["full_build_per_launch"] = c
    out = os.path.join(REPO, "gpurun_out", "r05_sweep_phase_budget_%s.json" % config)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps(rep["full_build_per_launch"]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--config", default="config2")
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        run(a.config)
