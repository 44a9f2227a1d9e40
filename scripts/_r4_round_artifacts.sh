#!/bin/bash
# round-4 artefacts beside the refreshed profile set: shards (default mode), step timeline, whole-genome fit, drop-in path
mkdir -p gpurun_out/r4
timeout -k 10 500 bash scripts/shards.sh > gpurun_out/r4/shards_exact.txt 2>&1
echo "shards done: $?"
timeout -k 10 200 bash scripts/step_timeline.sh > gpurun_out/r4/step_timeline.txt 2>&1
echo "timeline done: $?"
timeout -k 10 200 python3 scripts/fit_bench.py > gpurun_out/r4/fit_bench.json 2> gpurun_out/r4/fit.err
echo "fit done: $?"
timeout -k 10 200 python3 scripts/dropin_bench.py > gpurun_out/r4/dropin_bench.json 2> gpurun_out/r4/dropin.err
echo "dropin done: $?"
tail -12 gpurun_out/r4/shards_exact.txt | cut -c1-300
