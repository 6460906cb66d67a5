#!/bin/sh
# Ablation builds of the split tiles' production schedule in csrc/conv_gemm.hip (round 6): the product objects + conv_gemm.hip compiled with
# -DCG_ABL=<bits> (1 no activation split, 2 no DMA wait before the slice barrier, 4 no DMA in the K loop, 8 no epilogue, 16 no MFMAs; results
# are garbage, the TIME is the measurement).  Writes adafocus_amd/csrc/exp_build/libadafocus_hip_cg<bits>.so; run with ADAF_LIB=<that file>.
# usage: build_cg_abl.sh 1 2 4 ...
set -e
cd "$(dirname "$0")/../../adafocus_amd/csrc"
mkdir -p exp_build
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize -DCG_ABL=$k -c conv_gemm.hip -o exp_build/conv_gemm_abl$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o exp_build/conv_gemm_abl$k.o conv_lat.o crop.o misc_ops.o mobilenetv2.o mbconv.o gru_scan.o stem.o effnet.o mbconv_whole.o mbstrip.o -o exp_build/libadafocus_hip_cg$k.so
done
ls -la exp_build/*cg*.so
