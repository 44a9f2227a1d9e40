mkdir -p gpurun_out/r4
export TMPDIR=/tmp
STEPS=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/tr -o t -- python3 scripts/one_step.py > gpurun_out/r4/tr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name'][:60]
    print("%10.1f us  +%8.1f us  %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n))
PY
