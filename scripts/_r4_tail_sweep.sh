#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/tail_sweep.txt; : > $out
for v in 60,20 60,40 65,35 70,15 70,30 75,25 80,20 90,10 60,20; do
  echo "TAIL_PCT=$v" >> $out
  CONSENRICH_AMD_TAIL_PCT=$v CFGS=-1,-1,-1 timeout -k 10 120 python3 scripts/tune.py 2>&1 | cut -c1-120 >> $out
done
cat $out
