#!/bin/bash
# Kernel timeline of one bench step (rocprofv3 --kernel-trace of scripts/one_step.py), start times relative to the step's
# statistics kernel.  XTOL=2: the throughput mode; SHARD=8:6: an LPT shard; OUT: output directory (default gpurun_out/r5).
OUT=${OUT:-gpurun_out/r5}
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/tr
STEPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 scripts/one_step.py > $OUT/tr.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_stats' in r['Kernel_Name']]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
end = t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  +%8.1f us  (ends %8.1f)  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:90]))
    end = max(end, e)
print("step span (statistics of this step -> statistics of the next): %.1f us; last kernel ends at %.1f us" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3, (end - t0) / 1e3))
PY
