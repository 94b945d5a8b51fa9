#!/bin/bash
# Per-kernel times of a few denoiser passes (rocprofv3 --kernel-trace) -> gpurun_out/ks/<label>.csv
#   usage: scripts/kstats_unet.sh label [ENV=VAL ...]     (environment assignments are exported first)
set -eo pipefail
label=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
mkdir -p gpurun_out/ks
rm -rf /tmp/ks_$label
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/ks_$label -o p -- python3 scripts/prof_unet.py ${PROF_ARGS:-10240 4} > gpurun_out/ks/$label.log 2>&1
DB=$(find /tmp/ks_$label -name '*results.db' | head -n 1)
python3 scripts/rocpd_summary.py stats "$DB" | cut -c1-160 > gpurun_out/ks/$label.csv
python3 scripts/rocpd_summary.py seq "$DB" > gpurun_out/ks/$label.seq.csv
rm -rf /tmp/ks_$label
echo "$label done"
