set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in f32 split_bf16; do
for k in 0 32; do
  if [ $k = 0 ]; then unset ADAF_LIB; else export ADAF_LIB=$R/adafocus_amd/csrc/exp_build/libadafocus_hip_cg$k.so; fi
  if [ $m = f32 ]; then unset MATH; else export MATH=$m; fi
  timeout 300 python $R/tools/layer_table.py 96 1024 > /tmp/cg32_${m}_$k.txt 2>&1
done
done
python - <<'PY' > $OUT/r6_cg_abl32.txt
def rd(f):
    rows=[]
    for l in open(f):
        p=l.split()
        if len(p)>=6 and p[1].replace('.','').isdigit() and p[0]!="total": rows.append((p[0],float(p[1]),p[4]))
        if l.startswith("total"): rows.append(("total",float(p[1]),""))
    return rows
print("CG_ABL=32: every tile of the lean kernels reads the same 1024 activation rows (activations from L2, nothing else changed); ms per launch")
for m in ("f32","split_bf16"):
    a,b=rd("/tmp/cg32_%s_0.txt"%m),rd("/tmp/cg32_%s_32.txt"%m)
    print("== trunk arithmetic %s"%m)
    print("%-16s %4s %8s %8s %8s"%("launch","tile","product","abl32","saved"))
    for x,y in zip(a,b): print("%-16s %4s %8.4f %8.4f %8.4f"%(x[0],x[2],x[1],y[1],x[1]-y[1]))
PY
cat $OUT/r6_cg_abl32.txt
