"""Is the default mode still the SEQUENTIAL recursion of its own arithmetic on a very long chain?  One chain of N bins x M samples
(default: chr1 at 50 bp x 64, the chain on which round 6 found 11 % of the level values one float32 ulp off the oracle): the
default mode (superblocks, one launch), its pass form (CONSENRICH_AMD_SB_ASYNC=0), the sequential yardstick
(CONSENRICH_AMD_SEQ_STATE=1) and the CPU oracle, every filtered-state value compared.   N=... M=... python scripts/exact_vs_seq_probe.py"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np

import cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams

n, m = int(os.environ.get("N", "4979129")), int(os.environ.get("M", "64"))
seed = int(os.environ.get("SEED", "1234"))
F = np.asarray(cases.F_TREND, np.float32)
Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)


def run(env):
    for k in ("CONSENRICH_AMD_SEQ_STATE", "CONSENRICH_AMD_SB_ASYNC"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with DeviceBatch(0, x_tol_ulps=0) as b:
        b.configure(ModelParams(state_dim=2), m, [n])
        b.synthesize(seed)
        b.stats()
        sd, sn = b.forward(L.RETURN_NLL)
        b.export(L.EXPORT_FORWARD)
        out = {"xf": b.download(0, "xf"), "Pf": b.download(0, "Pf"), "D": b.download(0, "D"), "nll": sn[0], "stats": b.run_stats()}
        ins = b.download_inputs(0) if env.get("WANT_INPUTS") else None
    return out, ins


seq, ins = run({"CONSENRICH_AMD_SEQ_STATE": "1", "WANT_INPUTS": "1"})
os.environ.pop("WANT_INPUTS", None)
sb, _ = run({})
pf, _ = run({"CONSENRICH_AMD_SB_ASYNC": "0"})
for name, r in (("superblocks, single launch", sb), ("superblocks, pass form", pf)):
    dl = int(np.count_nonzero(r["xf"][:, 0] != seq["xf"][:, 0]))
    dt = int(np.count_nonzero(r["xf"][:, 1] != seq["xf"][:, 1]))
    print(f"{name} vs sequential kernel: level values differing {dl}, trend values differing {dt}, Pf {int(np.count_nonzero(r['Pf'] != seq['Pf']))}, "
          f"D {int(np.count_nonzero(r['D'] != seq['D']))}, nll {r['nll'] == seq['nll']}; reruns_x {r['stats']['reruns_x']} bailouts {r['stats']['sb_bailouts']}")
from oracle import oracle as orc

d_, v_ = ins
xf, Pf, pn = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32)
D = np.zeros(n, np.float32)
orc.cforwardPass(matrixData=d_, matrixPluginMuncInit=v_, matrixF=F, matrixQ0=Q0, intervalToBlockMap=np.zeros(n, np.int32), blockCount=1,
                 stateInit=0.0, stateCovarInit=1000.0, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn, vectorD=D, returnNLL=True)
for name, r in (("sequential kernel", seq), ("superblocks, single launch", sb)):
    dl = r["xf"][:, 0] != xf[:, 0]
    dt = r["xf"][:, 1] != xf[:, 1]
    first = int(np.argmax(dl)) if dl.any() else -1
    # episodes of level differences and their lengths
    edges = np.flatnonzero(np.diff(np.concatenate(([0], dl.view(np.int8), [0]))))
    lens = (edges[1::2] - edges[0::2]) if edges.size else np.zeros(0, int)
    print(f"{name} vs ORACLE: level values differing {int(dl.sum())} ({dl.mean():.3g}), trend {int(dt.sum())}, Pf {int(np.count_nonzero(r['Pf'] != Pf))}, "
          f"D {int(np.count_nonzero(r['D'] != D))}; first level difference at bin {first}; {lens.size} episodes, median length "
          f"{int(np.median(lens)) if lens.size else 0}, longest {int(lens.max()) if lens.size else 0}; max |level| {float(np.abs(xf[:, 0]).max()):.1f}")
    if first >= 0:
        k = first
        print(f"   around the first difference (bin {k}): oracle level {xf[k-1:k+2, 0]!r}, device {r['xf'][k-1:k+2, 0]!r}; oracle trend {xf[k-1:k+2, 1]!r}, device {r['xf'][k-1:k+2, 1]!r}")
