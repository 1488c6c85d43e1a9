#!/bin/bash
# The oracle's own CPU tests with the oracle built under AddressSanitizer + UBSan (the checker gets checked):
#   bash tests/tools/oracle_asan.sh
set -e
cd "$(dirname "$0")/../.."
make -s -C oracle asan
cp oracle/libsmx_oracle.so /tmp/libsmx_oracle_plain.so
trap 'cp /tmp/libsmx_oracle_plain.so oracle/libsmx_oracle.so; touch oracle/libsmx_oracle.so; rm -f oracle/libsmx_oracle_asan.so' EXIT
cp oracle/libsmx_oracle_asan.so oracle/libsmx_oracle.so
touch oracle/libsmx_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest -x -q \
  tests/test_oracle_known_answers.py tests/test_oracle_properties.py tests/test_nn_oracle.py tests/test_golden.py
