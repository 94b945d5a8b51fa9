"""Summarises rocprofv3's rocpd database (the default output format of this ROCm) as text/CSV.

    python scripts/rocpd_summary.py stats  gpurun_out/prof_r1/bench_results.db  > profiles/r1_bench_kernel_stats.csv
    python scripts/rocpd_summary.py pmc    gpurun_out/pmc_fetch/f_results.db    > profiles/r1_unet_pmc_FETCH_SIZE.csv

`stats` = what `--kernel-trace --stats` reports per kernel (calls, total/avg/min/max ns, percent).
`pmc`   = per-kernel sums / per-launch means of every collected counter.
"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","VGPR","AGPR","LDS","Scratch"')
    for r in rows:
        print('"%s",%d,%d,%.1f,%.3f,%d,%d,%d,%d,%d,%d' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / total,
                                                          r[4], r[5], r[6], r[7], r[8], r[9]))


def seq(path):
    """every dispatch in start order: short kernel name, grid, duration (for launch-by-launch A/B of two builds)"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else "0")
    rows = db.execute("select name, %s, duration, start from kernels order by start" % gx).fetchall()
    for i, r in enumerate(rows):
        print("%d,%s,%d,%d" % (i, r[0][:90].replace(",", ";"), r[1], r[2]))


def pmc(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name order by sum(value) desc").fetchall()
    print('"Kernel","Counter","Launches","Sum","PerLaunch","AvgDurationNs"')
    for r in rows:
        print('"%s","%s",%d,%.6g,%.6g,%.1f' % r)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "seq": seq}[sys.argv[1]](sys.argv[2])
