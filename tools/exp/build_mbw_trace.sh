#!/bin/sh
# Trace build of the EfficientNet kernels: mbconv_whole.hip with -DMBW_TRACE and effnet.hip with -DEF_TRACE (lane 0 of every wave stamps
# s_memtime at the phase boundaries into a buffer set through adaf_mbw_set_trace / adaf_ef_set_trace), linked with the product objects.  Writes
# adafocus_amd/csrc/libadafocus_hip_mbwtrace.so; read by tools/mbw_trace.py and tools/dw_trace.py (never the shipped library).
set -e
cd "$(dirname "$0")/../../adafocus_amd/csrc"
make -j4 >/dev/null
mkdir -p exp_build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize -DMBW_TRACE ${MBW_EXTRA} -c mbconv_whole.hip -o exp_build/mbconv_whole_trace$$.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize -DEF_TRACE ${MBW_EXTRA} -c effnet.hip -o exp_build/effnet_trace$$.o &
objs=""
for f in api conv_gemm conv_lat crop misc_ops mobilenetv2 mbconv gru_scan stem; do objs="$objs $f.o"; done
wait
objs="$objs exp_build/effnet_trace$$.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs exp_build/mbconv_whole_trace$$.o -o ${MBW_OUT:-libadafocus_hip_mbwtrace.so}
echo built ${MBW_OUT:-libadafocus_hip_mbwtrace.so}
