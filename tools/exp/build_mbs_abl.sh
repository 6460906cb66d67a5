#!/bin/sh
# Ablation builds of csrc/mbstrip.hip (round 6): the product objects + mbstrip.hip compiled with -DMBS_ABL=<bits> (1 expand / stem MFMA chains off,
# 2 project chains off, 4 tap FMAs off, 8 ring reads off, 16 pixel loads off; results are garbage, the TIME is the measurement).
# Writes adafocus_amd/csrc/exp_build/libadafocus_hip_abl<bits>.so; run with ADAF_LIB=<that file>.   usage: build_mbs_abl.sh 1 2 3 4 ...
set -e
cd "$(dirname "$0")/../../adafocus_amd/csrc"
mkdir -p exp_build
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -DMBS_ABL=$k -c mbstrip.hip -o exp_build/mbstrip_abl$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o conv_gemm.o conv_lat.o crop.o misc_ops.o mobilenetv2.o mbconv.o gru_scan.o stem.o effnet.o mbconv_whole.o exp_build/mbstrip_abl$k.o -o exp_build/libadafocus_hip_abl$k.so
done
ls -la exp_build/*.so
