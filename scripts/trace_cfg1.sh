#!/bin/bash
# Kernel trace of one DDPM step of cfg1 (320 trajectories, unguided): gpurun_out/cfg1/step.txt
set -eo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/cfg1
rm -rf /tmp/cfg1t
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/cfg1t -o p -- python3 bench.py --workload cfg1_unguided --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/cfg1/bench.json 2> gpurun_out/cfg1/err.log
DB=$(find /tmp/cfg1t -name '*results.db' | head -n 1)
python3 - "$DB" > gpurun_out/cfg1/step.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration, grid_x, workgroup_x from kernels order by start").fetchall()
mq = [i for i, r in enumerate(rows) if "map_query" in r[0]]
i0, i1 = mq[150], mq[151]            # one step in the middle of the second call (no CFG: one map_query per step)
prev_end = None
tot = 0
for r in rows[i0:i1]:
    n = r[0].replace("void trace::", "").replace("trace::", "").split("(")[0]
    gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-64s %7.1f us  gap %5.1f us  wgs %5d" % (n[:64], r[2] / 1e3, gap, r[3] // max(r[4], 1)))
    prev_end = r[1] + r[2]
    tot += r[2]
print("sum %.1f us, span %.1f us, launches %d" % (tot / 1e3, (rows[i1][1] - rows[i0][1]) / 1e3, i1 - i0))
PY
rm -rf /tmp/cfg1t
cat gpurun_out/cfg1/step.txt
