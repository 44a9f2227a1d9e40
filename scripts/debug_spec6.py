import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from test_gpu_parity import _run_batch
n_list = [5000]
seq = _run_batch(32 * 512, (0, 0, 0), 2, n_list, 4, 100)
os.environ["CONSENRICH_AMD_DEBUG"]="1"
spec = _run_batch(32, (0,0,0), 2, n_list, 4, 100)
for name in ("Pf","xf","pnoise","D","xs","Ps"):
    a, b = seq[(0, name)].astype(np.float64), spec[(0, name)].astype(np.float64)
    d = np.abs(a-b).reshape(a.shape[0], -1).max(axis=1)
    rows = np.nonzero(d>0)[0]
    print(name, "nbad rows", len(rows), "first", rows[:10], "maxdiff", d.max(), "median diff of bad", np.median(d[rows]) if len(rows) else 0)
    if name=="Pf":
        for r in rows[:6]: print("   row", r, "seq", seq[(0,name)][r].ravel(), "spec", spec[(0,name)][r].ravel())
