"""Print the kernel timeline of the last step from a rocprofv3 --kernel-trace rocpd database (scripts/one_step.py run)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_stats' in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]; prev_end = None; tot = 0
for r in rows[i0:i1 + 1]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0
    print("%-52s s%-3s start %8.1f us dur %7.1f gap %6.1f" % (r[0][:52], r[3], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap))
    prev_end = max(prev_end or 0, r[2]); tot += (r[2] - r[1]) / 1e3
print("kernel sum us", tot - (rows[i1][2] - rows[i1][1]) / 1e3, "step span", (rows[i1][1] - t0) / 1e3)
