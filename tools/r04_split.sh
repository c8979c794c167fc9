cd /tmp
for v in 0 1; do
IMMESH_SPLIT_GENERAL=$v timeout 60 python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 200 --warmup 5 --cpu-seconds 0 --extra-configs 0 --profile-scans 5 --nu-scans 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_scan']; print('SPLIT $v', d['value'], d['ms_per_step'], 'list', k.get('replay_list_kernel'), 'sub', k.get('replay_sub_kernel'), 'fused', k.get('replay_fused_kernel'))"
done
