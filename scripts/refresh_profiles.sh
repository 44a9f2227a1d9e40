#!/bin/bash
# Regenerate the round's profile artifacts on the GPU box (run through gpurun from the repo root; R = round tag, default r02):
#   gpurun_out/p/{R_bench.json, R_bench_under_rocprofv3.json, R_kernel_stats.csv, R_pmc_*.csv, R_pmc_traffic.json, R_pmc_sq*.json}
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE separately, no tracing flags); the program itself
# follows `--` (python3, no env / shell hop).
set -e
export TMPDIR=/tmp
R=${R:-r02}
O=gpurun_out/p
rm -rf $O && mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras > $O/${R}_bench_under_rocprofv3.json 2> $O/kt.err
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats.csv
export STEPS=3
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 scripts/one_step.py > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 scripts/one_step.py > $O/write.log 2>&1
cp "$(find $O/fetch -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_fetch_size.csv
cp "$(find $O/write -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_write_size.csv
python3 scripts/pmc_traffic.py $O/${R}_pmc_fetch_size.csv $O/${R}_pmc_write_size.csv $O/${R}_pmc_traffic.json > $O/${R}_pmc_traffic.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/sq -o s -- python3 scripts/one_step.py > $O/sq.log 2>&1
cp "$(find $O/sq -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_sq.csv
(cd scripts && python3 pmc_sq.py ../$O/${R}_pmc_sq.csv ../$O/${R}_pmc_sq.json) > $O/${R}_pmc_sq.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_e -o f -- python3 scripts/ecm_once.py > $O/fetch_e.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_e -o w -- python3 scripts/ecm_once.py > $O/write_e.log 2>&1
python3 scripts/pmc_traffic.py "$(find $O/fetch_e -name '*counter_collection.csv' | head -1)" "$(find $O/write_e -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_traffic_ecm.json > $O/${R}_pmc_traffic_ecm.txt
# the bit-exact mode (superblock state chain): kernel trace of the same steps with XTOL=0
export XTOL=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktx -o kt -- python3 scripts/one_step.py > $O/ktx.log 2>&1
cp "$(find $O/ktx -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_exact_mode.csv
unset XTOL
rm -rf $O/kt $O/ktx $O/fetch $O/write $O/sq $O/fetch_e $O/write_e
ls -la $O
