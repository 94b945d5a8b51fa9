#!/bin/bash
# ms per denoiser pass (forward + input-VJP) at several row counts for library variants: "label=path" arguments
for rows in 640 1280 2560 5120; do
  for spec in "$@"; do
    label=${spec%%=*}; lib=${spec#*=}
    r=$(TRACE_HIP_LIB=$lib timeout -k 10 120 python3 scripts/prof_unet.py $rows 30 2>/dev/null | grep "ms per pass") || exit 1
    echo "rows $rows $label: $r"
  done
done
