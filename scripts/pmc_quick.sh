#!/bin/bash
# HBM traffic of a bench step per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/pmc_traffic.py) in one
# validation mode: MODE=ulp2 (default) or exact.  Output: $OUT/pmc_traffic_$MODE.{json,txt}
set -e
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/r5}
MODE=${MODE:-ulp2}
mkdir -p $OUT
export STEPS=3
if [ $MODE = ulp2 ]; then export XTOL=2; else unset XTOL; fi
rm -rf $OUT/fetch $OUT/write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 scripts/one_step.py > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 scripts/one_step.py > $OUT/write.log 2>&1
python3 scripts/pmc_traffic.py "$(find $OUT/fetch -name '*counter_collection.csv' | head -1)" "$(find $OUT/write -name '*counter_collection.csv' | head -1)" $OUT/pmc_traffic_$MODE.json $STEPS > $OUT/pmc_traffic_$MODE.txt
rm -rf $OUT/fetch $OUT/write
cat $OUT/pmc_traffic_$MODE.txt
