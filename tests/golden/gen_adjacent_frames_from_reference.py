#!/usr/bin/env python3
"""Golden table of the reference's neighbour-view rule: get_adjacent_frames_idxs
(raynet/utils/training_utils.py:9-60, behind Scene._get_neighbor_idxs, common/scene.py:41-57),
produced by the reference's own function (lib2to3 scratch copy under /tmp, see
gen_pointcloud_from_reference.py for the loader).  Output: ref_adjacent_frames.json."""
import importlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_pointcloud_from_reference import REF, load_reference   # noqa: E402


def main():
    _, _, _, scratch = load_reference()
    pkg = os.path.join(scratch, "refpc")
    shutil.copy(os.path.join(REF, "raynet", "utils", "training_utils.py"),
                os.path.join(pkg, "utils", "training_utils.py"))
    import subprocess
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n",
                           os.path.join(pkg, "utils", "training_utils.py")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tu = importlib.import_module("refpc.utils.training_utils")
    table = []
    for n_frames in (5, 12, 50):
        for n_adjacent in (1, 2, 3, 4, 6):
            if n_adjacent >= n_frames:
                continue
            for skip in (0, 1):
                for ref_idx in range(n_frames + 1):      # n_frames itself is accepted (:23)
                    try:
                        out = [int(v) for v in tu.get_adjacent_frames_idxs(ref_idx, n_frames,
                                                                             n_adjacent, skip)]
                    except Exception as e:               # noqa: BLE001
                        out = "error:" + type(e).__name__
                    table.append([ref_idx, n_frames, n_adjacent, skip, out])
    json.dump(table, open(os.path.join(HERE, "ref_adjacent_frames.json"), "w"))
    print(len(table), "cases; e.g.", table[:3], table[-3:])
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
