#!/bin/bash
# A/B of library variants inside one gpurun call: ms per denoiser pass (forward + input-VJP, bench shape) for each
# "label=ENV=VAL,ENV=VAL" argument; stops at the first failure.
mkdir -p gpurun_out/ab
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
    timeout -k 10 120 python3 scripts/prof_unet.py ${PROF_ARGS:-10240 6} > gpurun_out/ab/$label.log 2>&1 ) || { echo "$label FAILED"; tail -5 gpurun_out/ab/$label.log; exit 1; }
  echo "$label: $(grep -E "ms per pass|^done" gpurun_out/ab/$label.log | tr "\n" " ")"
done
