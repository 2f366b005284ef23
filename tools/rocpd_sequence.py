#!/usr/bin/env python3
"""Time-ordered listing of ONE steady-state iteration from a rocprofv3 rocpd kernel trace.
usage: rocpd_sequence.py results.db MARK_KERNEL_SUBSTR [--list]
The window is from the end of the second-to-last launch of the mark kernel (e.g. adamw_kernel) to the end of its last launch.
Prints: kernel-time / gap totals, a histogram by launch duration, the time spent per family in the forward and the backward half
(split at the cfg_loss kernels when present), and with --list every launch (start offset, duration, gap before, grid, name)."""
import collections
import sqlite3
import sys


def main():
    db, mark = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = list(c.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    marks = [i for i, r in enumerate(rows) if mark in r[0]]
    assert len(marks) >= 2, f"{len(marks)} launches of *{mark}*"
    win = rows[marks[-2] + 1: marks[-1] + 1]
    t0 = rows[marks[-2]][2]
    wall = (win[-1][2] - t0) / 1e3
    ktime = sum(r[2] - r[1] for r in win) / 1e3
    gaps, prev = [], t0
    for r in win:
        gaps.append(max(0, r[1] - prev) / 1e3)
        prev = max(prev, r[2])
    print(f"# one iteration: {len(win)} launches, wall {wall:.1f} us, kernel time {ktime:.1f} us, idle between launches {sum(gaps):.1f} us "
          f"(mean gap {sum(gaps) / len(gaps):.2f} us)")
    edges = [0, 4, 6, 8, 10, 15, 20, 30, 50, 100, 1e9]
    hist = collections.OrderedDict()
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [r for r in win if lo <= (r[2] - r[1]) / 1e3 < hi]
        hist[f"{lo:g}-{hi:g} us"] = (len(sel), sum(r[2] - r[1] for r in sel) / 1e3)
    print("# launches by duration: " + "; ".join(f"{k}: {n} launches {t:.0f} us" for k, (n, t) in hist.items() if n))
    split = [i for i, r in enumerate(win) if "cfg_loss_fwd" in r[0]]
    halves = [("forward", win[:split[0]]), ("backward + optimiser", win[split[0]:])] if split else [("all", win)]

    def fam(n):
        for k in ("train_gemm_pair", "train_gemm_skinny", "train_gemm_direct", "train_gemm_kernel", "big_gemm_nt", "big_gemm_tn", "gn_bwd", "gn_fwd", "gn_",
                  "ln_bwd", "ln_fwd", "attn_small_bwd", "attn_small_fwd", "attn", "adamw", "repack", "at::native", "act_", "split2", "concat2"):
            if k in n:
                return k
        return "other"
    for name, part in halves:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in part:
            a = agg[fam(r[0])]
            a[0] += 1
            a[1] += (r[2] - r[1]) / 1e3
        span = (part[-1][2] - (part[0][1])) / 1e3 if part else 0.0
        print(f"# {name}: {len(part)} launches, span {span:.0f} us: " + ", ".join(f"{k} {v[0]}x {v[1]:.0f}us" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
    if "--list" in sys.argv:
        prev = t0
        for r, g in zip(win, gaps):
            print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:8.2f} {g:6.2f} {r[3]:8d} {r[4]:5d}  {r[0][:110]}")


if __name__ == "__main__":
    main()
