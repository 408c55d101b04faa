#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 rocpd sqlite file (--kernel-trace, optionally
--memory-copy-trace): every dispatch and copy of the last complete step in start order with its
duration and the idle gap before it, then the totals (busy, idle, wall).

    rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o tl -- python bench.py --steps 4 ...
    python tools/step_timeline.py /tmp/tl/.../tl_results.db [out.txt]

A step starts at a k_traverse dispatch (the first kernel of a pass)."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"\d+(k_[a-z_0-9]+?)(I[LN]|E)", name)      # mangled: ...N_111k_sweep_mapILi2E...
    if m:
        return m.group(1)
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    return name.split("(")[0][:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    ev = [(s, e, short(n), "kernel") for s, e, n in db.execute(
        "select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")]
    try:
        ev += [(s, e, "copy %s %d B" % (n, b), "copy") for s, e, n, b in db.execute(
            "select start, end, name, size from rocpd_memory_copy")]
    except sqlite3.Error:
        pass
    ev.sort()
    starts = [i for i, x in enumerate(ev) if x[2].startswith("k_traverse")]
    if len(starts) < 2:
        print("no complete step in the trace")
        return
    a, b = starts[-2], starts[-1]
    step = ev[a:b]
    t0 = step[0][0]
    out = ["%10s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "what")]
    busy_end = t0
    busy = 0
    for s, e, n, kind in step:
        gap = s - busy_end
        out.append("%10.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n))
        if e > busy_end:
            busy += e - max(s, busy_end)
            busy_end = e
    wall = ev[b][0] - t0
    out.append("step wall (traverse to traverse) %.1f us, busy (union) %.1f us, idle %.1f us, "
               "tail after the last event %.1f us" % (wall / 1e3, busy / 1e3, (wall - busy) / 1e3,
                                                     (ev[b][0] - busy_end) / 1e3))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
