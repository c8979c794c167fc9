#!/bin/bash
# HBM traffic per launch (rocprofv3 --pmc, one counter per pass, serial mode) with the per-dispatch values of the admission kernels listed: how many
# dispatches does the collection record, and how much do they differ?   usage: tools/r04_traffic.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; T=${1:-t}; shift; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$ctr
  ( time IMMESH_SERIAL_SAFE=1 timeout 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_$ctr -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 2 --profile-scans 0 --nu-scans 0 --async-mesh 0 "$@" > /tmp/rp_$ctr.log 2>&1 ) 2>&1 | grep real
  tail -2 /tmp/rp_$ctr.log | cut -c1-200
done
F=$(find /tmp/rp_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find /tmp/rp_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rows", len(rows), "dispatch ids", min(int(r["Dispatch_Id"]) for r in rows), "..", max(int(r["Dispatch_Id"]) for r in rows))
for name in ("mesh_append_prepare", "mesh_begin_scan", "residual_persistent", "mesh_knn"):
    v = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in rows if name in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print(name, len(v), [round(x[1] * 2 / 1024, 2) for x in v][:20], "MB read (2 x FETCH_SIZE KiB)")
PY
python $R/tools/pmc_traffic.py $F $W $O/traffic_$T.json
