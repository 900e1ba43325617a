#!/bin/bash
# probe builds of ctc.hip: tools/_probe/libctc_<name>.so  (usage: build_ctc_variants.sh "NAME:-DFLAG1 -DFLAG2" ...)
cd /root/repo/speechless_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c ctc.hip -o /tmp/ctc_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/_probe/libctc_$name.so capi.o conv_nt_bf16.o wgrad_tn_bf16.o conv_f32.o /tmp/ctc_$name.o misc.o spectrogram.o conv_chain_bf16.o ) &
done
wait
ls /root/repo/tools/_probe/
