#!/bin/bash
# A/B of two library builds on one box: swaps the package's libsrx.so, runs bench.py for each, restores
set -e
cp super-resolution_amd/libsrx.so /tmp/libsrx_keep.so
for round in 1 2; do
  for v in HEAD IL; do
    cp scratch/libsrx_$v.so super-resolution_amd/libsrx.so
    echo "$v $(timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile $@ 2>/dev/null | cut -c60-175)"
  done
done
cp /tmp/libsrx_keep.so super-resolution_amd/libsrx.so
