#!/bin/bash
# round 6: the pose chain's kernels alone (--mesh 0) and beside the mesher, 20-scan run, per-kernel means from the rocprofv3 timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
summ() { python - $1 <<'PY'
import sys, collections
acc=collections.defaultdict(list)
lines=open(sys.argv[1]).read().splitlines()
print(lines[0])
for ln in lines[1:]:
    p=ln.split(None,3)
    if len(p)==4:
        try: acc[p[3].strip().split('(')[0][:40]].append(float(p[1]))
        except ValueError: pass
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])): print(f"{k:42s} n={len(v):3d} mean {sum(v)/len(v):6.1f} us  max {max(v):6.1f}")
PY
}
bash $R/tools/timeline.sh 36 --nu-scans 0 --steps 20 --warmup 5 --mesh 0 > $O/tl_mesh0.txt 2>&1; summ $O/tl_mesh0.txt; cat $O/tl_mesh0.txt | head -40
