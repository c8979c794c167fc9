#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.txt.  Run on the GPU box (builds the binary there if it is missing).
[ -x ${GRAFT_REPO_ROOT:-/root/repo}/tools/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 ${GRAFT_REPO_ROOT:-/root/repo}/tools/fetch_calib.hip -o ${GRAFT_REPO_ROOT:-/root/repo}/tools/fetch_calib
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/fetch_calib.txt; : > $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$ctr
  timeout 100 rocprofv3 --pmc $ctr --output-format csv -d /tmp/fc_$ctr -- $R/tools/fetch_calib > /tmp/fc_$ctr.log 2>&1
  grep -E "expected per launch|write side" /tmp/fc_$ctr.log | head -2 >> $O
  f=$(find /tmp/fc_$ctr -name '*counter_collection.csv' | head -1)
  python - "$f" $ctr >> $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[2]:10s} {k:14s} launches {len(v)}  raw counter per launch: " + ", ".join(f"{x:.0f}" for x in v))
PY
done
cat $O
