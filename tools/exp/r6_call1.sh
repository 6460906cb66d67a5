set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_valu_overlap.hip -o /tmp/mvo 2>/dev/null && /tmp/mvo > gpurun_out/r6_mfma_valu_overlap.txt 2>&1
cat gpurun_out/r6_mfma_valu_overlap.txt
python tools/glancer_probe.py 1024 2>&1 | tail -1
python tools/glancer_probe.py 1024 5 2>&1 | tail -1
bash tools/profile_r6_glancer.sh r6base 2>&1 | tail -5
python bench.py > gpurun_out/r6base_bench.json 2> gpurun_out/r6base_bench.err; tail -c 1500 gpurun_out/r6base_bench.json
