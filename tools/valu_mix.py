#!/usr/bin/env python3
"""Static VALU instruction mix of the path's kernels, weighted with the MEASURED issue cost of
each instruction class (tools/valu_issue_bench.hip -> profiles/r05_valu_issue_bench.txt).  Since
round 5 bench.py no longer needs this estimate for the kernel it reports on: its counter pass
measures the VALUs' busy cycles per executed instruction directly (4 x SQ_ACTIVE_INST_VALU /
SQ_INSTS_VALU = 4.09 for k_sweep_map at config 2, against 3.80 from this static histogram); the
table stays as the fall-back for runs without counters and for the other kernels.

    python tools/valu_mix.py            # compiles the library to assembly (hipcc -S, ~40 s), writes
                                        # profiles/valu_issue.json

Classes (cycles per wave64 instruction and SIMD at 8 wavefronts per SIMD, nominal 2.4 GHz):
  fast   v_add/sub/mul/fma/fmac_f32, v_add/sub_u32, v_and/or/xor_b32        ~2.3 - 2.9
  std    everything else on the VALU: v_pk_*, *_dpp, v_max/min/med3, shifts, conversions,
         compares, moves, v_mad_u32_u24, v_mul_lo, bit counts, v_fma_f64, v_readlane      ~4.1
  trans  v_rcp / rsq / sqrt / exp / log / sin / cos _f32                                   ~8.1
The histogram is STATIC (every instruction of the kernel's text once, rarely taken fall-backs
included), so `mean_cycles` is an estimate of the dynamic mix, not a count of it; bench.py uses it
to turn SQ_INSTS_VALU into an issue time (valu_issue_ms) and says so."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd import _lib      # noqa: E402

# measured, profiles/r05_valu_issue_bench.txt (8 wavefronts per SIMD; r04: two runs on two boxes).
# v_cndmask_b32 with its mask in an SGPR pair issues like the rest of "std" (4.1 cycles); a stream of
# 1024 selects that all read VCC runs at 22.9 cycles each (round 4's 23.6: the same artefact of the
# benchmark's stream, not what a compare + select pair costs in a kernel -- the kernels' measured
# mean of 4.09 cycles per executed instruction leaves no room for it).
CYCLES = {"fast": 2.6, "std": 4.12, "trans": 8.12}
CYCLES_RANGE = {"fast": [2.2, 2.9], "std": [4.05, 4.85], "trans": [8.1, 8.15]}
FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac)_f32(_e32|_e64)?$|^v_(add|sub|subrev)_u32(_e32|_e64)?$|"
                  r"^v_(and|or|xor)_b32(_e32|_e64)?$")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag)?_f32")

# kernel name pattern -> label
KERNELS = {
    "sweep_map/config2": r"k_sweep_mapILi2ELi5ELi8ELi3ELb1EE",
    "sweep_map/config4": r"k_sweep_mapILi2ELi9ELi8ELi2ELb1EE",
    "bp/steady": r"k_bpILi6ELb1ELb0ELb1EE",
    "depth/steady": r"k_depthILi6ELb1ELb0ELb1EE",
    "scatter/box128x32": r"k_scatter_boxILb1ELi128ELi32ELb0EE",
    "traverse": r"k_traverseILb1EE",
}


def main():
    tmp = tempfile.mkdtemp(prefix="valu_mix_")
    asm = os.path.join(tmp, "rn.s")
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", asm,
                                                               os.path.join(_lib.CSRC, "raynet_hip.hip")],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")
    out = {"cycles_per_instruction": CYCLES, "cycles_range_over_runs": CYCLES_RANGE,
           "source": "tools/valu_issue_bench.hip on MI355X (profiles/r04_valu_issue_bench.txt), "
                     "static histograms by tools/valu_mix.py", "kernels": {}}
    for label, pat in KERNELS.items():
        rx = re.compile(r"^(_ZN\S*" + pat + r"\S*):")
        start = next((i for i, l in enumerate(text) if rx.match(l)), None)
        if start is None:
            continue
        hist = {"fast": 0, "std": 0, "trans": 0}
        for l in text[start + 1:]:
            t = l.strip()
            if t.startswith("s_endpgm"):
                break
            op = t.split()[0] if t and not t.startswith((";", ".")) else ""
            if not op.startswith("v_"):
                continue
            if "_dpp" in t.split(";")[0] and not op.startswith("v_mov"):
                cls = "std"
            elif TRANS.match(op):
                cls = "trans"
            elif FAST.match(op):
                cls = "fast"
            else:
                cls = "std"
            hist[cls] += 1
        n = sum(hist.values())
        out["kernels"][label] = dict(hist, static_valu=n, mean_cycles=round(
            sum(hist[c] * CYCLES[c] for c in hist) / max(n, 1), 3))
        print("%-20s %s  mean %.2f cycles" % (label, hist, out["kernels"][label]["mean_cycles"]))
    json.dump(out, open(os.path.join(REPO, "profiles", "valu_issue.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
