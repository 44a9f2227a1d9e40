#!/bin/bash
# Block length of the speculative chains, re-checked on the round's final kernels: ms per step by B (0 = the automatic choice),
# throughput mode and default mode, genome and the heaviest 1/8 shard.
export CFGS=-1,-1,-1
for mode in 2 0; do
  for shard in "" 8:6; do
    for B in 0 32 64 128 256; do
      echo -n "x_tol_ulps=$mode shard=${shard:-genome} B=$B  "
      CONSENRICH_AMD_XTOL_ULPS=$mode SHARD=$shard B=$B python3 scripts/tune.py | sed 's/reruns.*//'
    done
  done
done
