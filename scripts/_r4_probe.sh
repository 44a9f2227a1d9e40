#!/bin/bash
# round-4 probe: step time + state-chain debug counters for the current build
set -o pipefail
out=gpurun_out/r4/probe.txt
mkdir -p gpurun_out/r4
: > $out
run() {
  echo "== $1" >> $out
  CFGS=-1,-1,-1 timeout -k 10 200 python3 scripts/tune.py >> $out 2>&1
  CONSENRICH_AMD_SB_DEBUG=1 STEPS=2 timeout -k 10 100 python3 scripts/one_step.py 2>&1 | grep "csr\]" | grep -v "chains final" | tail -n 4 >> $out
}
run "default"
export CONSENRICH_AMD_TAIL_SPLIT=0
run "TAIL_SPLIT=0"
cat $out
