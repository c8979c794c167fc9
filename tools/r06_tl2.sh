#!/bin/bash
# round 6: kernel timelines (tools/timeline.sh) of the driver's short run and of the 500-scan steady state, per-kernel means of the window
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
env "$@" bash $R/tools/timeline.sh 200 --nu-scans 0 --steps 20 --warmup 5 > $O/tl_short.txt 2>&1; summ $O/tl_short.txt
env "$@" bash $R/tools/timeline.sh 200 --nu-scans 0 --gpu-scans 1 --steps 500 --warmup 20 > $O/tl_long.txt 2>&1; summ $O/tl_long.txt
