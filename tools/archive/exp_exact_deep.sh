cd $GRAFT_REPO_ROOT
for wl in config3 config3pyr; do for v in nodeep base nodeep base; do
 if [ $v != base ]; then export HSM_LIB=$PWD/hector_slam_amd/lib/variants/libhector_mi355_$v.so; else unset HSM_LIB; fi
 python bench.py --workload $wl --no-cpu --no-pmc --no-pyramid --no-pipelined --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exact_parity']
print('$wl $v exact kernel_us', round(e['kernel_ms']*1e3,1), 'parity', e.get('parity_vs_cpu',{}).get('bit_identical_pose_fraction'), 'fast', round(d['roofline']['kernel_ms']*1e3,1))"
done; done
