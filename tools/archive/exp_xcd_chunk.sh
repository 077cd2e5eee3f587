cd $GRAFT_REPO_ROOT
for wl in config3 config4; do for c in 0 16 0 16; do
 HSM_XCD_CHUNK=$c python bench.py --workload $wl --no-cpu --no-pmc --no-pyramid --no-pipelined --no-exact --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl chunk $c kernel_us', round(d['roofline']['kernel_ms']*1e3,1))"
done; done
