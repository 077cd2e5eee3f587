for c in 0 1 2 4 8 16 32; do
  HSM_XCD_CHUNK_EXACT=$c HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu --sustain-s 0 --no-exact 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c', round(d['value']/1e6,1),'M it/s kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
