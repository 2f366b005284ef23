#!/usr/bin/env python3
"""How much of a rocprofv3 kernel trace (rocpd database) ran concurrently: sum of kernel durations, time with >= 1 and >= 2 kernels
in flight, dispatches per queue.  Tuning tool for the forked branches of the replayed training pass.

    python tools/rocpd_overlap.py trace.db [--between KERNEL_SUBSTR]
"""
import argparse
import collections
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--between", default=None, help="only dispatches between the 2nd and the last launch of this kernel")
ap.add_argument("--timeline", type=int, default=0, help="print this many dispatches (start offset us, duration us, queue, name) from "
                                                        "the --skip'th dispatch that is not on the busiest queue")
ap.add_argument("--skip", type=int, default=100)
a = ap.parse_args()
c = sqlite3.connect(a.db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = ", ".join(x for x in ("d.start", "d.end", f"s.{name_col}", f"d.{qcol}" if qcol else "0", f"d.{scol}" if scol else "0"))
rows = list(c.execute(f"select {sel} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
if a.between:
    marks = [r[1] for r in rows if a.between in r[2]]
    lo, hi = marks[1], marks[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"# window: {len(marks) - 2} iterations, {(hi - lo) / 1e6:.3f} ms")
ev = []
for st, en, *_ in rows:
    ev.append((st, 1))
    ev.append((en, -1))
ev.sort()
depth, last, busy1, busy2 = 0, None, 0, 0
for t, d in ev:
    if last is not None:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
    depth += d
    last = t
tot = sum(r[1] - r[0] for r in rows)
print(f"dispatches {len(rows)}  sum of durations {tot / 1e6:.3f} ms  >=1 in flight {busy1 / 1e6:.3f} ms  >=2 in flight {busy2 / 1e6:.3f} ms")
for label, idx in (("queue", 3), ("stream", 4)):
    cnt = collections.Counter(r[idx] for r in rows)
    dur = collections.Counter()
    for r in rows:
        dur[r[idx]] += r[1] - r[0]
    print(f"# by {label}: " + ", ".join(f"{k}: {n} dispatches {dur[k] / 1e6:.2f} ms" for k, n in cnt.most_common(8)))

if a.timeline:
    main_q = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
    side = [i for i, r in enumerate(rows) if r[3] != main_q]
    if side:
        i0 = max(0, side[min(a.skip, len(side) - 1)] - 10)
        t0 = rows[i0][0]
        for st, en, name, q, _ in rows[i0:i0 + a.timeline]:
            print(f"{(st - t0) / 1e3:9.2f} {(en - st) / 1e3:7.2f} q{q} {'   ' * (0 if q == main_q else 1)}{name[:60]}")
