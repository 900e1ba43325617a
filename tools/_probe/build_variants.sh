#!/bin/bash
# probe builds of the chain kernel: tools/_probe/lib_<name>.so   (usage: build_variants.sh NAME...)
cd /root/repo/speechless_amd/csrc
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DSL_CHAIN_PROBE_$v -c conv_chain_bf16.hip -o /tmp/chain_$v.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/_probe/lib_$v.so capi.o conv_nt_bf16.o wgrad_tn_bf16.o conv_f32.o ctc.o misc.o spectrogram.o /tmp/chain_$v.o ) &
done
wait
ls /root/repo/tools/_probe/
