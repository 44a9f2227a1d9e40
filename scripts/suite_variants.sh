#!/bin/bash
# The GPU suite under the library's mode switches, one line per variant (run through gpurun from the repo root); the names of
# failing tests follow their variant's line.  VARIANTS="A=1 B=0" runs a subset.
ALL="CONSENRICH_AMD_WARMSTART=0 CONSENRICH_AMD_SEQ_STATE=1 CONSENRICH_AMD_DEFER=0 CONSENRICH_AMD_DMA=0 CONSENRICH_AMD_SB_BINS=4096 \
 CONSENRICH_AMD_SB_ASYNC=0 CONSENRICH_AMD_SB_SPIN_LIMIT=1 CONSENRICH_AMD_TAIL_SPLIT=0 CONSENRICH_AMD_TAIL_PCT=5,5 CONSENRICH_AMD_NATIN=0"
for v in ${VARIANTS:-$ALL}; do
  out=$(env $v python3 -m pytest tests -m gpu -q 2>&1)
  echo "$v: $(echo "$out" | tail -n 1)"
  echo "$out" | grep -E "^(FAILED|E   )" | head -n 40 || true
done
