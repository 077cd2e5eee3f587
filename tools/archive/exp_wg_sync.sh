cd $GRAFT_REPO_ROOT
for c in 0 1 0 1; do
 HSM_WG_SYNC=$c python bench.py --workload config4 --no-cpu --no-pmc --no-pyramid --no-pipelined --no-exact --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 wg_sync $c kernel_us', round(d['roofline']['kernel_ms']*1e3,1))"
done
