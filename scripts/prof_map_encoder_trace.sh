#!/bin/bash
# rocprofv3 kernel trace of the HIP map encoder at 1024 images -> gpurun_out/me/map_encoder_kernel_stats.csv
set -eo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/me
rm -rf /tmp/me_trace
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/me_trace -o p -- python3 scripts/prof_map_encoder.py 1024 2 > gpurun_out/me/trace_run.log 2>&1
DB=$(find /tmp/me_trace -name '*results.db' | head -n 1)
python3 scripts/rocpd_summary.py stats "$DB" > gpurun_out/me/map_encoder_kernel_stats.csv
rm -rf /tmp/me_trace
echo "trace done"
