#!/bin/bash
# ab.sh ROUNDS VARIANT... : bench.py (config 2, no profile) with each scratch/variants/libsrx_VARIANT.so, alternating
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  SRX_LIB_PATH=$PWD/scratch/variants/libsrx_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 30 $BENCH_ARGS 2>/dev/null | python scratch/ms.py $v
done; done
