#!/bin/bash
# Where do the lone wavefronts of the per-scan kernels spend their cycles?  SQ counters per dispatch, averaged per kernel (serial mode, few scans).
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves); INSTS_* are instruction counts summed over waves.
# usage (GPU box): tools/pmc_insts.sh > gpurun_out/pmc_insts.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-seconds 0 --steps 6 --warmup 2 --profile-scans 0 --async-mesh 0 --extra-configs 0"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pi_$tag
  IMMESH_SERIAL_SAFE=1 timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pi_$tag -- python $R/bench.py $ARGS > /tmp/pi_$tag.log 2>&1
  f=$(find /tmp/pi_$tag -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
first = min((int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"].startswith("residual")), default=0)
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    if int(r["Dispatch_Id"]) < first: continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
print("%-36s %6s " % ("kernel", "calls") + " ".join("%14s" % c[-14:] for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get(names[1] if len(names) > 1 else names[0], [0]))):
    n = max(len(v) for v in d.values())
    print("%-36s %6d " % (k, n) + " ".join("%14.0f" % (sum(d.get(c, [0])) / max(1, len(d.get(c, [0])))) for c in names))
PY
done
