for v in base early0 base early0; do
  if [ $v = base ]; then unset HSM_LIB; else export HSM_LIB=$PWD/hector_slam_amd/lib/variants/libhector_mi355_$v.so; fi
  for w in config4 config3pyr; do
    HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --workload $w --compact --no-pmc --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $w', round(d['value']/1e6,1),'M it/s', round(d['ms_per_step']*1e3,1),'us/step kernel', d['roofline'].get('kernel_ms'))"
  done
done
