#!/bin/bash
# builds tools/latency_c.c against include/omx.h + the product library and runs it (GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=${OMX_HIP_LIBDIR:-$ROOT/openmeters_amd/csrc}
gcc -O2 -std=c99 -Wall -I "$ROOT/include" "$ROOT/tools/latency_c.c" -o /tmp/omx_latency_c -L "$LIB" -lomx_hip -lm -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined
/tmp/omx_latency_c
