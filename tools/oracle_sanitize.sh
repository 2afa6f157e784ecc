#!/bin/bash
# The oracle (test infrastructure) under UBSan and ASan on the CPU — GPU AddressSanitizer is not available on this pool.  Builds two instrumented
# copies of oracle/libwloracle.so under /tmp, runs the known-answer and golden-vector tests against each, and restores the normal build.
# (ASan is preloaded into an uninstrumented python: the one test that makes the oracle THROW — the multigrid size check — is deselected there,
#  libasan's __cxa_throw interceptor cannot resolve the real symbol in that setup.)
set -e
cd "$(dirname "$0")/.."
cp oracle/libwloracle.so /tmp/libwloracle_orig.so
restore() { cp /tmp/libwloracle_orig.so oracle/libwloracle.so; touch oracle/libwloracle.so oracle/libwloracle_omp.so; }
trap restore EXIT
F="-O1 -g -std=c++17 -fPIC -march=x86-64-v3 -ffp-contract=off -shared"
g++ $F -fsanitize=undefined -fno-sanitize-recover=undefined -o /tmp/libwloracle_ubsan.so oracle/wl_oracle_capi.cpp
g++ $F -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/libwloracle_asan.so oracle/wl_oracle_capi.cpp
cp /tmp/libwloracle_ubsan.so oracle/libwloracle.so; touch oracle/libwloracle.so
echo "== UBSan"; UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_oracle_kat.py tests/test_golden.py -q -m "not gpu" -p no:cacheprovider | tail -1
cp /tmp/libwloracle_asan.so oracle/libwloracle.so; touch oracle/libwloracle.so
echo "== ASan"; ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(g++ -print-file-name=libasan.so) python -m pytest tests/test_oracle_kat.py tests/test_golden.py -q -m "not gpu" -p no:cacheprovider --deselect tests/test_oracle_kat.py::test_multigrid_index_maps | tail -1
