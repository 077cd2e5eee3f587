for v in 0 1 0 1; do
  HSM_EXACT_SPEC1=$v HSM_BENCH_DETAILS=/tmp/b.json timeout 600 python bench.py --workload config2 --compact --no-pmc --no-cpu > /dev/null 2>&1
  python -c "import json; d=json.load(open('/tmp/b.json')); print('spec1=$v', 'hsm_match us', d['latency_us'], d['config']['kernel'].get('kernel'), 'node loop', json.dumps(d.get('node_loop_cpp'))[:200], 'cycle', d.get('slam_cycle_us'))"
done
