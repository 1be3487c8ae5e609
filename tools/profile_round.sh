#!/bin/bash
# profile_round.sh NAME: rocprofv3 kernel statistics of the default bench command and the memory-side PMC
# passes (one pass per counter group), summarised into gpurun_out/profiles_NAME/ (copy what is to be kept into profiles/).  Run on the GPU box from the repo root.
set -e
R=$(pwd); N=$1; OUT=$R/gpurun_out/profiles_$N; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$N -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --extras 0 > $R/gpurun_out/prof_$N.log 2>&1
cp $(ls $R/gpurun_out/prof_$N/*kernel_stats.csv $R/gpurun_out/prof_$N/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${N}_bench_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$R/gpurun_out/pmc_${N}_$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-profile --extras 0 --steps 2 --warmup 1 > $d.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py $OUT/${N}_pmc_traffic.json gpurun_out/pmc_${N}_FETCH_SIZE gpurun_out/pmc_${N}_WRITE_SIZE gpurun_out/pmc_${N}_TCC_HIT_sum
rm -rf gpurun_out/prof_$N gpurun_out/pmc_${N}_*
grep '^{' gpurun_out/prof_$N.log | tail -1
