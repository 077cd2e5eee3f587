#!/bin/bash
# One sanitizer pass over the CPU side (race/sanitizer row): the two CPU checkers and the reference
# node-loop driver are rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle SAN=1 -> build/san,
# _ref/san) and the whole CPU test suite runs against them (libstdc++ is preloaded next to the ASan runtime: the node checker
# throws and catches a C++ exception, and ASan's __cxa_throw interceptor must find the real one when it initialises) (python itself is not instrumented: the ASan runtime is
# preloaded, leak detection off because the interpreter never frees at exit).  Output: $1 (default gpurun_out/sanitizer_cpu.txt; copy it to profiles/rNN/)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -C oracle SAN=1 > /dev/null || exit 1
OUT=${1:-gpurun_out/sanitizer_cpu.txt}
{
  echo "# $(date -u +%F) g++ $(g++ -dumpversion): -fsanitize=address,undefined on hector_oracle.cpp, ref_shim.cpp (reference headers), slam_driver.cpp"
  HSM_ORACLE_SAN=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libstdc++.so.6)" \
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | grep -v "^$"
  echo "# sanitizer reports in the output above: $(grep -c -E 'runtime error|ERROR: AddressSanitizer' "$OUT.tmp" 2>/dev/null || echo 0)"
} > "$OUT.tmp" 2>&1
n=$(grep -c -E 'runtime error|ERROR: AddressSanitizer' "$OUT.tmp")
sed -i "s/^# sanitizer reports in the output above:.*/# sanitizer reports (runtime error | AddressSanitizer): $n/" "$OUT.tmp"
mv "$OUT.tmp" "$OUT"
tail -5 "$OUT"
