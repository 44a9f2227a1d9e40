#!/bin/bash
# Step and ECM iteration of the LPT shards an N-GPU run would give each rank, emulated on ONE GPU (heaviest / lightest rank of
# 8, rank 0 of 4 and 2), then the whole genome on the same box.  Output -> profiles/<round>_shards.txt.
set -e
for s in 8:0 8:6 4:0 2:0; do
  echo "SHARD=$s"
  SHARD=$s CFGS=-1,-1,-1 python3 scripts/tune.py
  SHARD=$s ITERS=6 CPU=0 PROFILE=0 python3 scripts/ecm_bench.py
done
echo GENOME
CFGS=-1,-1,-1 python3 scripts/tune.py
ITERS=6 CPU=0 PROFILE=0 python3 scripts/ecm_bench.py
