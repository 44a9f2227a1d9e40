#!/bin/bash
# hunt for the SB_BINS=4096 flake in the context it showed up in: the whole GPU suite, N times; full text of any failure is kept
mkdir -p gpurun_out/r4
: > gpurun_out/r4/flake.txt
for i in $(seq 1 ${N:-4}); do
  CONSENRICH_AMD_SB_BINS=4096 timeout -k 10 400 python -m pytest tests -m gpu -q -x > gpurun_out/r4/flake_run.txt 2>&1
  tail -1 gpurun_out/r4/flake_run.txt | tee -a gpurun_out/r4/flake.txt
  if grep -q "failed" gpurun_out/r4/flake_run.txt; then cp gpurun_out/r4/flake_run.txt gpurun_out/r4/flake_fail_$i.txt; fi
done
