#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: rocpd_stats.py results.db [--top N] [--dispatches KERNEL_SUBSTR] [--between KERNEL_SUBSTR]
--between: only the dispatches from the end of the SECOND launch of the named kernel to the end of its last one (steady-state
iterations of a loop that ends every iteration with that kernel, e.g. adamw_kernel: capture / warm-up launches are left out)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    where = ""
    if "--between" in sys.argv:
        sub = sys.argv[sys.argv.index("--between") + 1]
        marks = [r[0] for r in c.execute(f"select d.end from {kd} d join {ks} s on d.kernel_id = s.id where s.{name_col} like ? order by d.start", (f"%{sub}%",))]
        assert len(marks) >= 3, f"--between {sub}: {len(marks)} launches"
        where = f"where d.start > {marks[1]} and d.end <= {marks[-1]}"
        print(f"# window: {len(marks) - 2} iterations between launches 2 and {len(marks)} of *{sub}* ({(marks[-1] - marks[1]) / 1e6 / (len(marks) - 2):.3f} ms per iteration wall)")
    q = f"""select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from {kd} d join {ks} s on d.kernel_id = s.id {where} group by s.{name_col} order by 3 desc"""
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:100]:100s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f}")
    print(f"TOTAL kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if "--dispatches" in sys.argv:
        sub = sys.argv[sys.argv.index("--dispatches") + 1]
        gcols = [x for x in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x", "lds_block_size", "group_segment_size") if x in cols]
        q = f"select s.{name_col}, d.end - d.start, {', '.join('d.' + g for g in gcols)} from {kd} d join {ks} s on d.kernel_id = s.id where s.{name_col} like ? order by d.start"
        for r in list(c.execute(q, (f"%{sub}%",)))[:400]:
            print(r[0][:60], f"{r[1] / 1e3:8.2f}us", dict(zip(gcols, r[2:])))


if __name__ == "__main__":
    main()
