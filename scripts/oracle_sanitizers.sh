#!/bin/bash
# Runs the CPU test suite against sanitizer builds of the C restatement (oracle/fa_oracle.c): UndefinedBehaviorSanitizer, then
# AddressSanitizer (preloaded into python; the one test that makes the reference's C++ build throw is left out — libasan's throw
# interceptor cannot resolve __cxa_throw when libstdc++ arrives later through dlopen).  Restores the normal build afterwards.
set -u
cd "$(dirname "$0")/.." || exit 1
cp oracle/libfa_oracle.so /tmp/libfa_oracle.so.keep
trap 'cp /tmp/libfa_oracle.so.keep oracle/libfa_oracle.so' EXIT
rc=0
gcc -O1 -g -ffp-contract=off -fPIC -shared -std=c11 -fsanitize=undefined -o oracle/libfa_oracle.so oracle/fa_oracle.c -lm || exit 1
UBSAN_OPTIONS=print_stacktrace=1 python -m pytest tests/ -q -m "not gpu" -p no:cacheprovider > /tmp/oracle_ubsan.log 2>&1 || rc=1
if grep -q "runtime error" /tmp/oracle_ubsan.log; then grep "runtime error" /tmp/oracle_ubsan.log | sort | uniq -c | head; rc=1; fi
echo "UBSan: $(tail -1 /tmp/oracle_ubsan.log)"
gcc -O1 -g -ffp-contract=off -fPIC -shared -std=c11 -fsanitize=address -o oracle/libfa_oracle.so oracle/fa_oracle.c -lm || exit 1
ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/oracle_asanrep LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/ -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_oracle_ahc.py::test_status_contract_of_reference_build --deselect tests/test_text_fuzz.py::test_text_entries_under_address_and_ub_sanitizers \
    --deselect tests/test_abi.py::test_every_entry_survives_null_and_zero_arguments > /tmp/oracle_asan.log 2>&1 || rc=1
ls /tmp/oracle_asanrep.* > /dev/null 2>&1 && { head -20 /tmp/oracle_asanrep.*; rc=1; }
echo "ASan:  $(tail -1 /tmp/oracle_asan.log)"
exit $rc
