#!/bin/bash
# Builds a variant of libtrace_hip.so from unet.hip compiled with extra flags (A/B runs inside one gpurun call):
#   scripts/build_variant.sh NAME -DTRACE_BF3_NST=3 ...   ->  trace_amd/lib/var_NAME.so   (run with TRACE_HIP_LIB=that path)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p trace_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I include -I trace_amd/csrc \
  -fno-slp-vectorize "$@" -c trace_amd/csrc/unet.hip -o trace_amd/lib/obj/unet_var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o trace_amd/lib/var_$name.so trace_amd/lib/obj/unet_var_$name.o \
  trace_amd/lib/obj/steps.o trace_amd/lib/obj/guidance.o trace_amd/lib/obj/conv2d.o
rm -f trace_amd/lib/obj/unet_var_$name.o
echo trace_amd/lib/var_$name.so
