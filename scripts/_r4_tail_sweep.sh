#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/tail_sweep.txt; : > $out
for v in 50,15 40,15 30,15 30,10 20,10 20,5 10,10 60,20; do
  echo "TAIL_PCT=$v" >> $out
  CONSENRICH_AMD_TAIL_PCT=$v CFGS=-1,-1,-1 timeout -k 10 120 python3 scripts/tune.py 2>&1 | cut -c1-120 >> $out
done
cat $out
