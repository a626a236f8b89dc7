#!/bin/bash
# per-replica residency of launch n and n+1 (deterministic trajectories: two runs that differ by one timed launch), compact state
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in 20 21 25; do
AWSEM_HIP_LIBRARY=$R/scratch/libawsem_WALL.so AWSEM_PROF_DUMP=$R/gpurun_out/wg_persist_$n.txt timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps $n --folded > /dev/null 2>&1 || exit 3
done
python - <<'PY'
import numpy as np
def load(n):
    rows=[l.split() for l in open(f"gpurun_out/wg_persist_{n}.txt")]
    a=np.array([[int(r[0]),int(r[1]),int(r[2])] for r in rows]); a=a[np.argsort(a[:,0])]
    return (a[:,2]-a[:,1])*0.01
a,b,c=load(20),load(21),load(25)
print("residency mean/sd:",a.mean(),a.std(),b.mean(),b.std())
print("corr(n,n+1) %.3f corr(n,n+5) %.3f"%(np.corrcoef(a,b)[0,1],np.corrcoef(a,c)[0,1]))
for x in (a,b):
    g=x.reshape(4,256)
    print("within-CU sd %.1f  CU-mean sd %.1f  max %.1f  mean of CU max %.1f"%(np.sqrt(((g-g.mean(0))**2).mean()), g.mean(0).std(), x.max(), g.max(0).mean()))
d=(a.reshape(4,256)-a.reshape(4,256).mean(0)).ravel(); e=(b.reshape(4,256)-b.reshape(4,256).mean(0)).ravel()
print("corr of within-CU deviation n vs n+1: %.3f"%np.corrcoef(d,e)[0,1])
PY
