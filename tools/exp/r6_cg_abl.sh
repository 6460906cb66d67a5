set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in ${ABLS:-0 1 2 4 6 7 8 16}; do
  if [ $k = 0 ]; then unset ADAF_LIB; else export ADAF_LIB=$R/adafocus_amd/csrc/exp_build/libadafocus_hip_cg$k.so; fi
  MATH=split_bf16 timeout 300 python $R/tools/layer_table.py 96 1024 > /tmp/cg_$k.txt 2>&1
done
python - <<'PY' > $OUT/r6_cg_abl.txt
import os
ks=[k for k in os.environ.get("ABLS","0 1 2 4 6 7 8 16").split()]
tab={}
names=[]
for k in ks:
    rows=[]
    for l in open("/tmp/cg_%s.txt"%k):
        p=l.split()
        if len(p)>=6 and p[1].replace('.','').isdigit() and p[0]!="total": rows.append((p[0],float(p[1]),p[4]))
        if l.startswith("total"): rows.append(("total",float(p[1]),""))
    tab[k]=rows
    if not names: names=[(r[0],r[2]) for r in rows]
print("CG_ABL: 1 no activation split, 2 no DMA wait, 4 no DMA in the K loop, 8 no epilogue, 16 no MFMAs (ms per launch, split_bf16 plan)")
print("%-16s %4s "%("launch","tile")+" ".join("%7s"%("abl"+k) for k in ks))
for i,(n,t) in enumerate(names):
    print("%-16s %4s "%(n,t)+" ".join("%7.4f"%tab[k][i][1] if i<len(tab[k]) else "      -" for k in ks))
PY
cat $OUT/r6_cg_abl.txt
