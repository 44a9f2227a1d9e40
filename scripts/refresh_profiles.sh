#!/bin/bash
# Regenerate the round's profile artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/p/{r01_bench.json, r01_bench_under_rocprofv3.json, r01_kernel_stats.csv, r01_pmc_*.csv, r01_pmc_traffic.json}
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE separately, no tracing flags).
set -e
export TMPDIR=/tmp
O=gpurun_out/p
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/r01_bench.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/r01_bench_under_rocprofv3.json 2> $O/kt.err
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/r01_kernel_stats.csv
STEPS=3 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 scripts/one_step.py > $O/fetch.log 2>&1
STEPS=3 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 scripts/one_step.py > $O/write.log 2>&1
cp "$(find $O/fetch -name '*counter_collection.csv' | head -1)" $O/r01_pmc_fetch_size.csv
cp "$(find $O/write -name '*counter_collection.csv' | head -1)" $O/r01_pmc_write_size.csv
python3 scripts/pmc_traffic.py $O/r01_pmc_fetch_size.csv $O/r01_pmc_write_size.csv $O/r01_pmc_traffic.json
rm -rf $O/kt $O/fetch $O/write
ls -la $O
