#!/bin/bash
# Superblock length of the bit-exact state chain on SMALL batches (one chain; one 1/8-genome shard): ms per step by
# CONSENRICH_AMD_SB_BINS (default: chosen from the batch, ensure_sb_view).  Output: one line per setting.
export CFGS=-1,-1,-1
for bins in default 1024 2048 4096 8192 16384; do
  if [ $bins = default ]; then unset CONSENRICH_AMD_SB_BINS; else export CONSENRICH_AMD_SB_BINS=$bins; fi
  echo -n "c2 (1e6 x 4, forward only) SB_BINS=$bins  "
  python3 bench.py --config c2 --no-cpu-baseline --no-extras --steps 20 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms/step %.3f  state chain %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))'
  echo -n "shard 8:6 (chr2-sized + ...) SB_BINS=$bins  "
  SHARD=8:6 python3 scripts/tune.py | sed 's/reruns.*//'
  echo -n "genome SB_BINS=$bins  "
  python3 scripts/tune.py | sed 's/reruns.*//'
done
