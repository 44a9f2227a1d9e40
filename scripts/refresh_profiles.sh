#!/bin/bash
# Regenerate the round's profile artifacts on the GPU box (run through gpurun from the repo root; R = round tag, default r03):
#   gpurun_out/p/{R_bench.json, R_bench_under_rocprofv3.json, R_kernel_stats*.csv, R_pmc_*.csv, R_pmc_traffic*.json, R_pmc_sq*.json}
# The default validation mode is the bit-exact one (the bench line's `value`); XTOL=2 selects the opt-in throughput mode.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE separately, no tracing flags); the program itself
# follows `--` (python3, no env / shell hop).
set -e
export TMPDIR=/tmp
R=${R:-r04}
O=gpurun_out/p
rm -rf $O && mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras > $O/${R}_bench_under_rocprofv3.json 2> $O/kt.err
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats.csv
export STEPS=3
for mode in exact ulp2; do
  if [ $mode = ulp2 ]; then export XTOL=2; S=""; else unset XTOL; S="_exact"; fi
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 scripts/one_step.py > $O/fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 scripts/one_step.py > $O/write.log 2>&1
  cp "$(find $O/fetch -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_fetch_size${S}.csv
  cp "$(find $O/write -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_write_size${S}.csv
  python3 scripts/pmc_traffic.py $O/${R}_pmc_fetch_size${S}.csv $O/${R}_pmc_write_size${S}.csv $O/${R}_pmc_traffic${S}.json $STEPS > $O/${R}_pmc_traffic${S}.txt
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/sq -o s -- python3 scripts/one_step.py > $O/sq.log 2>&1
  cp "$(find $O/sq -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_sq${S}.csv
  (cd scripts && python3 pmc_sq.py ../$O/${R}_pmc_sq${S}.csv ../$O/${R}_pmc_sq${S}.json) > $O/${R}_pmc_sq${S}.txt
  rm -rf $O/fetch $O/write $O/sq
done
# kernel trace of the throughput-mode steps (the bench line's `throughput_mode`)
export XTOL=2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktx -o kt -- python3 scripts/one_step.py > $O/ktx.log 2>&1
cp "$(find $O/ktx -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_throughput_mode.csv
unset XTOL
rm -rf $O/kt $O/ktx
ls -la $O
