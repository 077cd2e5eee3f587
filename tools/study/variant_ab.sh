#!/bin/bash
# A/B of library variants built by tools/build_variants.py (HSM_VARIANTS="base name1 name2 ..."): the driver bench line per variant
cd "${GRAFT_REPO_ROOT:-.}"
for v in ${HSM_VARIANTS:-base}; do
  export HSM_LIB=$PWD/hector_slam_amd/lib/variants/libhector_mi355_$v.so
  HSM_BENCH_DETAILS=/tmp/b.json timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu --sustain-s 0 --no-exact 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']/1e6,1),'M it/s kernel', round(r['kernel_ms']*1e3,2), 'sclk', r.get('sclk_hz'), 'frac@clk', r.get('frac_at_measured_clock'))"
done
