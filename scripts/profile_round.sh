#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench command (one timed call, no CPU baseline leg)
#   2. separate --pmc passes over the denoiser forward + input-VJP at the bench shape:
#      FETCH_SIZE, WRITE_SIZE (HBM traffic), and an SQ/GRBM pass (MFMA busy, LDS conflicts, clock)
# Every step has its own time limit and the chain stops at the first failure.
#   usage: bash scripts/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"

timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/stats.err"
DB=$(find "$OUT/stats" -name '*results.db' | head -n 1)
python3 scripts/rocpd_summary.py stats "$DB" > "$OUT/bench_kernel_stats.csv"
python3 scripts/step_trace.py "$DB" 50 > "$OUT/one_ddpm_step_launches.txt" || true
echo "stats done"

for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o p -- python3 scripts/prof_unet.py 10240 2 bwd nosplit > "$OUT/pmc_$C.log" 2>&1
  DB=$(find "$OUT/pmc_$C" -name '*results.db' | head -n 1)
  python3 scripts/rocpd_summary.py pmc "$DB" > "$OUT/unet_pmc_$C.csv"
  echo "$C done"
done

timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_SQ" -o p -- python3 scripts/prof_unet.py 10240 2 bwd nosplit > "$OUT/pmc_SQ.log" 2>&1
DB=$(find "$OUT/pmc_SQ" -name '*results.db' | head -n 1)
python3 scripts/rocpd_summary.py pmc "$DB" > "$OUT/unet_pmc_SQ.csv"
echo "SQ done"
# the databases are large: keep only the summaries
rm -rf "$OUT/stats" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ
ls -la "$OUT"
