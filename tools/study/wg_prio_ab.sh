for v in p3 p3f p3s9 p3j2 p3 p3f; do
  export HSM_LIB=$PWD/hector_slam_amd/lib/variants/libhector_mi355_$v.so
  HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu --sustain-s 0 --no-exact 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1),'M it/s kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
