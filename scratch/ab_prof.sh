#!/bin/bash
set -e
cp super-resolution_amd/libsrx.so /tmp/libsrx_keep.so
for v in HEAD IL; do
  cp scratch/libsrx_$v.so super-resolution_amd/libsrx.so
  echo "$v $(timeout -k 10 200 python bench.py --no-cpu-baseline --wgrad-overlap 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_breakdown_ms'])")"
done
cp /tmp/libsrx_keep.so super-resolution_amd/libsrx.so
