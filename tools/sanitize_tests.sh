#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# The native reader / writer tests under AddressSanitizer + UBSan, then under ThreadSanitizer.
#   tools/sanitize_tests.sh [asan|tsan|both]      (default both; CPU only, ~10 min)
# Builds trtools_amd/libtrk_{asan,tsan}.so (csrc/Makefile), preloads the sanitizer runtime into python and points
# the package at the instrumented library (TRK_LIBTRK).  Leak checking is off: CPython never frees everything.
set -u
cd "$(dirname "$0")/.."
which=${1:-both}
TESTS="tests/test_vcfnative.py tests/test_vcfnative_hook.py tests/test_vcfnative_fuzz.py tests/test_vcf_writer_native.py tests/test_vcf_shards.py tests/test_tabix.py tests/test_batch_pipelines.py tests/test_harmonizer_and_flags.py tests/test_bgzf_native.py"
rc=0
run() {   # $1 = asan|tsan, $2 = runtime library name
  make -C trtools_amd/csrc "$1" || exit 2
  local rt; rt=$(g++ -print-file-name="$2")
  [ "$1" = asan ] && rt="$rt:$(g++ -print-file-name=libubsan.so)"
  local cxx; cxx=$(g++ -print-file-name=libstdc++.so.6)
  echo "== $1: $rt"
  LD_PRELOAD="$rt:$cxx" TRK_LIBTRK="$PWD/trtools_amd/libtrk_$1.so" \
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66 suppressions=$PWD/tools/tsan.supp" \
    TRK_VCF_READ_AHEAD=${TRK_VCF_READ_AHEAD:-1} \
    python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  local r=${PIPESTATUS[0]}
  [ "$r" -ne 0 ] && rc=$r
}
[ "$which" = asan ] || [ "$which" = both ] && run asan libasan.so
[ "$which" = tsan ] || [ "$which" = both ] && run tsan libtsan.so
exit $rc
