#!/bin/bash
# The other bench lines kept under profiles/ (R = round tag): BASELINE configs 3 and 5 on one GPU, the long-memory regime.
set -e
R=${R:-r04}
O=gpurun_out/p
mkdir -p $O
python3 bench.py --samples 8 --no-cpu-baseline > $O/${R}_bench_c3_hg38_200bp_x8.json
python3 bench.py --samples 64 --bin-bp 50 --no-cpu-baseline > $O/${R}_bench_c5_hg38_50bp_x64.json
python3 bench.py --q0 1e-3,1e-4 --no-cpu-baseline > $O/${R}_bench_q0_1e-3_1e-4.json
python3 bench.py --q0 1e-5,1e-6 --no-cpu-baseline > $O/${R}_bench_q0_1e-5_1e-6.json
python3 bench.py --q0 1e-6,1e-7 --no-cpu-baseline > $O/${R}_bench_q0_1e-6_1e-7.json
