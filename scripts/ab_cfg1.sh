#!/bin/bash
# A/B of small-launch settings on cfg1 (320 trajectories): "label=ENV=VAL,ENV=VAL" arguments, ms per 100-step call
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
    timeout -k 10 200 python3 bench.py --workload cfg1_unguided --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],1), 'ms', round(d['value']))" ) || exit 1
done
