"""Per-launch table of ONE DDPM step from a rocprofv3 rocpd database (kernel trace of bench.py):
    python scripts/step_trace.py gpurun_out/prof_b/bench_results.db [step_index]
Prints the kernel sequence between two consecutive map_query launches with durations."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, duration, grid_x, workgroup_x from kernels order by start").fetchall()
mq = [i for i, r in enumerate(rows) if "map_query" in r[0]]
# with CFG there are two map_query launches per step: a step starts at every second one
i0, i1 = mq[2 * which], mq[2 * which + 2]
tot = 0
agg = {}
for r in rows[i0:i1]:
    n = r[0].replace("void trace::", "").replace("trace::", "").split("(")[0]
    print("%-52s %8.1f us  wgs %6d" % (n[:52], r[2] / 1e3, r[3] // max(r[4], 1)))
    tot += r[2]
    key = n.split("<")[0]
    agg[key] = agg.get(key, 0) + r[2]
print("sum %.1f us, span %.1f us" % (tot / 1e3, (rows[i1][1] - rows[i0][1]) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("  %-40s %8.1f us %5.1f%%" % (k[:40], v / 1e3, 100.0 * v / tot))
