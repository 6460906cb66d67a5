set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/layer_table.py 128 512 0 > /tmp/t0.txt 2>&1
python $R/tools/layer_table.py 128 512 8 > /tmp/t8.txt 2>&1
paste -d'|' <(grep -v amdgpu /tmp/t0.txt | cut -c1-60) <(grep -v amdgpu /tmp/t8.txt | cut -c1-60) > $OUT/r6_tsm_c1.txt
cat $OUT/r6_tsm_c1.txt
