"""Derives the committed roofline evidence from the PMC summaries of scripts/profile_round.sh:

    python scripts/summarise_profiles.py gpurun_out/prof_r1c profiles r1

  * profiles/hbm_traffic.json     (read by bench.py for roofline.traffic): conv kernels' HBM bytes per launch =
    2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section), separate passes
  * profiles/<tag>_unet_mfma_util.txt: MFMA pipe utilisation per conv instantiation from the SQ pass
and copies the per-kernel CSVs / the bench line under the round's tag.
"""
import csv
import json
import os
import shutil
import subprocess
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[(r["Kernel"], r["Counter"])] = (int(r["Launches"]), float(r["Sum"]), float(r["PerLaunch"]),
                                          float(r["AvgDurationNs"]))
    return d


for name in ("bench_kernel_stats.csv", "bench_line_under_rocprof.json", "one_ddpm_step_launches.txt",
             "unet_pmc_FETCH_SIZE.csv", "unet_pmc_WRITE_SIZE.csv", "unet_pmc_SQ.csv"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
F, W, S = (load(os.path.join(src, "unet_pmc_%s.csv" % c)) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ"))
tf = tw = n = dur = 0
for (k, c), (l, s, p, d) in F.items():
    if "conv_" in k:
        tf += s
        n += l
        dur += d * l
for (k, c), (l, s, p, d) in W.items():
    if "conv_" in k:
        tw += s
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
out = {"kernel": "conv_bf3_kernel + conv_gemm_kernel (all instantiations)",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 scripts/prof_unet.py 10240 2  (separate passes, "
                  "scripts/profile_round.sh; denoiser fwd + input-VJP at the bench shape, 10240 rows)",
       "launches": n, "fetch_size_kb_per_launch_raw": tf / n, "write_size_kb_per_launch": tw / n,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for coalesced streaming reads -> doubled "
                     "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "hbm_bytes_per_launch": (2 * tf + tw) / n * 1024, "avg_duration_ns": dur / n, "commit": commit}
json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
rows, tot_busy, tot_avail = [], 0.0, 0.0
for k in sorted(set(k[0] for k in S)):
    if "conv_" not in k:
        continue
    g = lambda c: S[(k, c)]
    L, durns = g("SQ_WAVE_CYCLES")[0], g("SQ_WAVE_CYCLES")[3]
    gui = g("GRBM_GUI_ACTIVE")[2] / 8.0                     # per-XCD cycles per launch
    busy, avail = g("SQ_VALU_MFMA_BUSY_CYCLES")[2], gui * 1024       # 256 CUs x 4 SIMDs
    wc = g("SQ_WAVE_CYCLES")[2]
    rows.append((k.replace("void trace::", "").replace("(trace::ConvArgs)", ""), L, durns / 1e3,
                 gui / (durns * 1e-9) / 1e9, busy / avail,
                 g("SQ_LDS_BANK_CONFLICT")[2] / max(g("SQ_LDS_IDX_ACTIVE")[2], 1),
                 g("SQ_WAIT_ANY")[2] / wc, g("SQ_WAIT_INST_ANY")[2] / wc, g("SQ_ACTIVE_INST_ANY")[2] / wc))
    tot_busy += busy * L
    tot_avail += avail * L
with open(os.path.join(dst, "%s_unet_mfma_util.txt" % tag), "w") as f:
    f.write("# from %s_unet_pmc_SQ.csv: MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 "
            "SIMDs), clock = GRBM_GUI_ACTIVE/8/duration\n" % tag)
    f.write("%-44s %5s %8s %6s %9s %9s %8s %9s %7s\n" % ("kernel", "calls", "avg_us", "GHz", "mfma_util", "lds_confl",
                                                         "wait_any", "wait_inst", "active"))
    for r in sorted(rows, key=lambda r: -r[1] * r[2]):
        f.write("%-44s %5d %8.1f %6.2f %9.3f %9.3f %8.3f %9.3f %7.3f\n" % r)
    f.write("all conv launches, time-weighted: mfma_util %.3f\n" % (tot_busy / tot_avail))
print(json.dumps(out))
print(open(os.path.join(dst, "%s_unet_mfma_util.txt" % tag)).read())
