#!/usr/bin/env python3
"""Reduce the lines a -DTSDF_PHASE_TIMER=1 build of libtsdf_hip.so appends to $TSDF_HIP_PHASE_FILE (one per k_integrate
launch: the shader-clock cycles every wave spent between the phase marks of its row loop, summed over the launch's waves)
to shares of a wave's lifetime and cycles per wave-row.

usage: phase_reduce.py FILE [--skip N]     (--skip: leading launches to drop, e.g. the first-after-reset launch and warm-up)
"""
import json
import sys
from collections import defaultdict

NAMES = {
    0: "loop control + LDS reads + transform + projection + certificate (+ exact fp64 fallback)",
    1: "early voxel loads + frame gather ISSUED",
    2: "wait: gathered depths (and early voxel words) arrive",
    3: "raw distances, observed / in-band tests",
    4: "normalise ladder (in-band rows), late loads issued",
    5: "wait: late-asked voxel words arrive",
    6: "decode, flags, hinge rest test, count / colour / distance update",
    7: "select, change detection, stores issued",
    9: "block prologue: tables, row transforms, flags, barrier",
    10: "loop exit",
    12: "epilogue: barrier + flag write-out",
}


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    groups = defaultdict(list)
    for i, ln in enumerate(open(path)):
        d = json.loads(ln)
        groups[(d["color"], d["count"], d["allin"], d["live"], d["packed"], d["blocks"])].append(d["phase"])
    for key, rows in groups.items():
        rows = rows[skip:] if len(rows) > skip else rows
        n = len(rows)
        ph = [sum(r[k] for r in rows) / n for k in range(16)]
        waves = key[5] * 4
        total = ph[15]
        print(f"# colour={key[0]} counting={key[1]} allin={key[2]} live={key[3]} packed={key[4]} blocks={key[5]} ({waves} waves), "
              f"{n} launches averaged")
        print(f"# wave lifetime {total / waves:,.0f} cycles on average; observed rows per wave {ph[8] / waves:.2f}")
        print(f"{'phase':>5}  {'share':>6}  {'cycles/wave':>12}  what")
        acc = 0.0
        for k in sorted(NAMES):
            acc += ph[k]
            print(f"{k:>5}  {ph[k] / total:6.1%}  {ph[k] / waves:12,.0f}  {NAMES[k]}")
        print(f"{'sum':>5}  {acc / total:6.1%}  (the rest: the marks' own scalar-memory round trips after the last mark of a path)")
        print()


if __name__ == "__main__":
    main()
