#!/bin/bash
# ab.sh ROUNDS VARIANT... : bench.py (config 2, no profile) with each tools/variants/libsrx_VARIANT.so, alternating
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  SRX_LIB_PATH=$PWD/tools/variants/libsrx_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 30 $BENCH_ARGS 2>/dev/null | python tools/ms.py $v
done; done
