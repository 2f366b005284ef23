#!/usr/bin/env python3
"""Break one replayed denoiser step of a rocprofv3 rocpd trace down by (kernel, workgroups, LDS bytes)."""
import collections
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
rows = list(c.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.group_segment_size from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# a replayed step is delimited by its tail launch (round 6: jen1_step_tail; the next step's head work is inside it), else by pack_input
anchor = 'step_tail' if sum('step_tail' in r[0] for r in rows) > which + 1 else 'pack_input'
idx = [i + 1 for i, r in enumerate(rows) if anchor in r[0]] if anchor == 'step_tail' else [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = idx[which], idx[which + 1]
step = rows[a:b]
print("launches in step:", len(step), "span us:", (step[-1][2] - step[0][1]) / 1e3, "sum kernel us:", sum(r[2] - r[1] for r in step) / 1e3)
gaps = [(step[i + 1][1] - step[i][2]) / 1e3 for i in range(len(step) - 1)]
print("gap avg/median/max us:", statistics.mean(gaps), statistics.median(gaps), max(gaps))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r[0]
    key = ('stream_gemm' if 'stream_gemm' in n else 'tile_gemm' if ('tile_gemm' in n or 'TileArgs' in n) else 'conv_gemm(wide)' if 'conv_gemm' in n else 'norm_apply' if 'norm_apply' in n else 'attention' if 'attention' in n else n.split('(')[0][-40:])
    wg = (r[3] // r[6]) * r[4] * r[5]
    k = (key, wg, r[7])
    agg[k][0] += 1
    agg[k][1] += (r[2] - r[1]) / 1e3
print("%-42s %8s %8s %5s %9s %8s" % ("kernel", "WGs", "lds", "n", "total_us", "avg_us"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-42s %8d %8d %5d %9.1f %8.2f" % (k[0][:42], k[1], k[2], v[0], v[1], v[1] / v[0]))
