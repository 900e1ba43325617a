#!/bin/bash
# Probe builds of ONE kernel file, linked against the other objects of the last regular build:
#   tools/build_probe.sh conv_chain_bf16.hip "TIMES:-DSL_CHAIN_PROBE_TIMES" "NOMFMA:-DSL_CHAIN_PROBE_NO_MFMA"
#   tools/build_probe.sh ctc.hip "CLK:-DSL_PROBE_CTC_CLOCK -fno-slp-vectorize"
# -> tools/_probe/lib_<NAME>.so (git-ignored; travels to the GPU box), used through SL_LIB_PATH=tools/_probe/lib_<NAME>.so
# (wrong results by construction for the timing probes: see tools/README.md)
set -e
SRC=$1; shift
cd "$(dirname "$0")/../speechless_amd/csrc"
mkdir -p ../../tools/_probe
OBJS=""
for f in capi conv_nt_bf16 wgrad_tn_bf16 conv_f32 ctc misc spectrogram conv_chain_bf16 conv1x1_bwd_bf16 split3; do
  [ "$f.hip" = "$SRC" ] || OBJS="$OBJS $f.o"
done
OBJS="$OBJS conv_nt_f16.o wgrad_tn_f16.o"  # (the -DSL_ELEM_F16 translation units of the last regular build, as they are)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $SRC -o /tmp/probe_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_probe/lib_$name.so $OBJS /tmp/probe_$name.o ) &
done
wait
ls ../../tools/_probe/
