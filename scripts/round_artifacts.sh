#!/bin/bash
# The measurement artefacts of a round, regenerated on ONE GPU box from the repo root (run through gpurun; R = round tag).  Output:
# gpurun_out/p/${R}_*; copy what is to be judged into profiles/.  PART selects a subset (bench pmc shards misc), default all.
#   bench : one bench line per BASELINE config (c2 forward only, c3, c4 = the headline, c5), each with roofline + cpu_baseline and both
#           validation modes; the same commands under rocprofv3 --kernel-trace --stats (kernel_stats_*.csv)
#   pmc   : HBM traffic (FETCH_SIZE x 2, WRITE_SIZE: separate passes) and the SQ breakdown of the c4 step, both modes
#   shards: step and ECM iteration of the LPT shards of a 2 / 4 / 8-GPU run emulated on one GPU, both modes
#   misc  : step timelines, ECM kernel shares, whole-genome fit, per-call drop-in entries, exact-mode fuzz
export TMPDIR=/tmp
R=${R:-r06}
O=gpurun_out/p
PART=${PART:-"bench pmc shards misc"}
mkdir -p $O
for part in $PART; do
case $part in
bench)
  for c in c4 c2 c3 c5; do
    timeout -k 10 600 python3 bench.py --config $c > $O/${R}_bench_$c.json 2> $O/bench_$c.err || echo "bench $c FAILED"
    rm -rf $O/kt
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --config $c --no-cpu-baseline --no-extras > $O/${R}_bench_${c}_under_rocprofv3.json 2> $O/kt_$c.err || echo "rocprof bench $c FAILED"
    cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_$c.csv
    rm -rf $O/kt
    echo "bench $c done"
  done
  cp $O/${R}_bench_c4.json $O/${R}_bench.json
  # long-memory process noise (the floor the reference's Q0 seed clamps to is 1e-6): windows lengthen, repair runs multiply
  for q in 1e-5,1e-6 1e-6,1e-7; do
    timeout -k 10 600 python3 bench.py --config c4 --q0 $q --no-cpu-baseline > $O/${R}_bench_q0_$q.json 2> $O/bench_q0_$q.err || echo "bench q0 $q FAILED"
  done
  # the throughput-mode steps of c4 under the kernel trace (the bench line's `throughput_mode`)
  XTOL=2 STEPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktx -o kt -- python3 scripts/one_step.py > $O/ktx.log 2>&1
  cp "$(find $O/ktx -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_c4_throughput_mode.csv
  rm -rf $O/ktx
  ;;
pmc)
  export STEPS=3
  for mode in exact ulp2; do
    if [ $mode = ulp2 ]; then export XTOL=2; S=""; else unset XTOL; S="_exact"; fi
    rm -rf $O/fetch $O/write $O/sq
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 scripts/one_step.py > $O/fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 scripts/one_step.py > $O/write.log 2>&1
    cp "$(find $O/fetch -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_fetch_size${S}.csv
    cp "$(find $O/write -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_write_size${S}.csv
    python3 scripts/pmc_traffic.py $O/${R}_pmc_fetch_size${S}.csv $O/${R}_pmc_write_size${S}.csv $O/${R}_pmc_traffic${S}.json $STEPS > $O/${R}_pmc_traffic${S}.txt
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/sq -o s -- python3 scripts/one_step.py > $O/sq.log 2>&1
    cp "$(find $O/sq -name '*counter_collection.csv' | head -1)" $O/${R}_pmc_sq${S}.csv
    (cd scripts && python3 pmc_sq.py ../$O/${R}_pmc_sq${S}.csv ../$O/${R}_pmc_sq${S}.json) > $O/${R}_pmc_sq${S}.txt
    rm -rf $O/fetch $O/write $O/sq
    echo "pmc $mode done"
  done
  unset XTOL
  ;;
shards)
  timeout -k 10 600 bash scripts/shards.sh > $O/${R}_shards_exact_mode.txt 2>&1; echo "shards exact: $?"
  CONSENRICH_AMD_XTOL_ULPS=2 timeout -k 10 500 bash scripts/shards.sh > $O/${R}_shards_throughput_mode.txt 2>&1; echo "shards ulp2: $?"
  ;;
misc)
  OUT=$O/tl timeout -k 10 200 bash scripts/step_timeline.sh > $O/${R}_step_timeline.txt 2>&1
  XTOL=2 OUT=$O/tl timeout -k 10 200 bash scripts/step_timeline.sh > $O/${R}_step_timeline_throughput_mode.txt 2>&1
  XTOL=2 SHARD=8:6 OUT=$O/tl timeout -k 10 200 bash scripts/step_timeline.sh > $O/${R}_step_timeline_throughput_mode_shard8.txt 2>&1
  rm -rf $O/tl
  ITERS=3 CPU=0 PROFILE=1 timeout -k 10 300 python3 scripts/ecm_bench.py > $O/${R}_ecm_kernels.txt 2>&1
  PROFILE=0 timeout -k 10 300 python3 scripts/fit_bench.py > $O/${R}_fit_bench_plain.json 2> $O/fit.err; echo "fit: $?"
  PROFILE=0 DIAG=1 timeout -k 10 300 python3 scripts/fit_bench.py > $O/${R}_fit_bench_with_phase_diagnostics.json 2>> $O/fit.err; echo "fit diag: $?"
  OUTER=8 timeout -k 10 600 python3 scripts/dropin_bench.py > $O/${R}_dropin_bench_outer8.json 2> $O/dropin.err; echo "dropin 8: $?"
  OUTER=32 NO_SINGLE=1 timeout -k 10 900 python3 scripts/dropin_bench.py > $O/${R}_dropin_bench.json 2>> $O/dropin.err; echo "dropin 32: $?"
  TRIALS=32 timeout -k 10 600 python3 scripts/fuzz_exact.py > $O/${R}_fuzz_exact.txt 2>&1; echo "fuzz: $?"
  ;;
esac
done
ls $O | head -80
