cd /tmp && export TMPDIR=/tmp
IMMESH_DEBUG=1 timeout 200 python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh 0 2>/tmp/dbg.err | grep '^{' | cut -c1-100
grep -E '^\[append_prepare' /tmp/dbg.err | tail -4
grep -E '^\[mesh\]' /tmp/dbg.err | tail -2
