#!/bin/bash
# One rocprofv3 --pmc pass over the denoiser (forward + input-VJP at the bench shape) -> gpurun_out/pmc/<label>.csv
#   usage: scripts/pmc_pass.sh label COUNTER [COUNTER ...]     (at most 8 SQ_ counters per pass)
set -eo pipefail
label=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rm -rf /tmp/pmc_$label
timeout -s KILL 150 rocprofv3 --pmc "$@" -d /tmp/pmc_$label -o p -- python3 scripts/prof_unet.py ${PROF_ARGS:-10240 2} > gpurun_out/pmc/$label.log 2>&1
DB=$(find /tmp/pmc_$label -name '*results.db' | head -n 1)
python3 scripts/rocpd_summary.py pmc "$DB" > gpurun_out/pmc/$label.csv
rm -rf /tmp/pmc_$label
echo "$label done"
