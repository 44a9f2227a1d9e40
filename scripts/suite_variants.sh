#!/bin/bash
# The GPU suite under the library's mode switches, one line per variant (run through gpurun from the repo root).
for v in CONSENRICH_AMD_WARMSTART=0 CONSENRICH_AMD_WS_MAX_BLOCK=256 CONSENRICH_AMD_SEQ_STATE=1 CONSENRICH_AMD_SB_STATE=0 \
         CONSENRICH_AMD_FOLD_CHECK=0 CONSENRICH_AMD_DEFER=0 CONSENRICH_AMD_DEFER_ITER=0 CONSENRICH_AMD_UNITF=0 CONSENRICH_AMD_FUSE=0 \
         CONSENRICH_AMD_NATOUT_D=0 CONSENRICH_AMD_STATS_WIDE=0 CONSENRICH_AMD_PREFAULT_THREADS=0; do
  r=$(env $v python3 -m pytest tests -m gpu -q 2>&1 | tail -1)
  echo "$v: $r"
done
