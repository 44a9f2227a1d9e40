#!/bin/bash
# Sweeps of the bit-exact state chain's tuning on the bench workload: superblock length (SWEEP=bins) or the delta-form
# repair pass's fallback rule (SWEEP=adv, "min bins per round, from round").  Output: one tune.py line per setting.
for v in ${VALUES}; do
  echo "$SWEEP=$v"
  if [ "$SWEEP" = bins ]; then export CONSENRICH_AMD_SB_BINS=$v; else export CONSENRICH_AMD_SB_ADV=$v; fi
  CFGS=-1,-1,-1 python3 scripts/tune.py
  CONSENRICH_AMD_SB_DEBUG=1 STEPS=1 python3 scripts/one_step.py 2>&1 | grep "csr\]" | tail -n 2
done
