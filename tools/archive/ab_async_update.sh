timeout 1300 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
show() { python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$1', {k:(v if not isinstance(v,dict) else {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a in ('median','p90')}) for k,v in d.items() if k in ('latency_us','update_latency_us','update_complete_us','slam_cycle_us','ms_per_step','match_ms','update_ms','value')})
"; }
for z in 4096 0; do HSM_UPDATE_ZEROCOPY_MAX=$z python bench.py --workload config2 --no-cpu 2>/dev/null | show "zc=$z"; done
HSM_ASYNC_UPDATE=0 python bench.py --workload config2 --no-cpu 2>/dev/null | show "sync"
python bench.py --workload config5 --no-cpu 2>/dev/null | show "c5 async"
HSM_ASYNC_UPDATE=0 python bench.py --workload config5 --no-cpu 2>/dev/null | show "c5 sync"
