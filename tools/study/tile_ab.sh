for v in base tile base tile; do
  if [ $v = base ]; then unset HSM_LIB; else export HSM_LIB=$PWD/hector_slam_amd/lib/variants/libhector_mi355_$v.so; fi
  HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --sustain-s 0 --no-exact 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v headline', round(d['value']/1e6,1),'M it/s kernel', round(d['roofline']['kernel_ms']*1e3,2), 'bit-identical', d.get('cpu_baseline',{}).get('bit_identical_pose_fraction'))"
  HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --workload config4 --compact --no-pmc --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config4', round(d['value']/1e6,1),'M it/s', round(d['ms_per_step']*1e3,1),'us/step')"
done
