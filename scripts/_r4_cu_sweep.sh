#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/cu_split.txt; : > $out
for cfg in "0 50,15" "0 60,40" "160 50,15" "160 30,10" "160 20,5" "176 30,10" "192 30,10" "192 15,5" "0 60,40"; do
  set -- $cfg
  echo "CU_SPLIT=$1 TAIL_PCT=$2" >> $out
  CONSENRICH_AMD_CU_SPLIT=$1 CONSENRICH_AMD_TAIL_PCT=$2 CFGS=-1,-1,-1 timeout -k 10 120 python3 scripts/tune.py 2>&1 | cut -c1-400 >> $out
done
cat $out | cut -c1-330
