"""TEST INFRASTRUCTURE ONLY -- the reference's bedGraph writer.

The reference writes every track with one pandas call (/root/reference/src/consenrich/consenrich.py:9797-9805):
``df[["Chromosome", "Start", "End", col]].to_csv(path, sep="\\t", header=False, index=False, mode=..., float_format="%.4f",
lineterminator="\\n")``.  pandas is installed in this image, so the oracle IS that call (pinned by construction);
``bedgraph_bytes_python`` is an independent statement of the same format with Python's own correctly rounded ``%``
formatting, used to cross-check.  Value transforms of the caller: ``round4`` = core.getPrimaryState (core.py:6145-6166,
np.round(x, 4) on float32), ``sqrt`` = uncertainty track (consenrich.py:9476-9477).  Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import io

import numpy as np


def transform_values(values, transform=None):
    v = np.ascontiguousarray(values, dtype=np.float32).copy()
    if transform == "round4":
        np.round(v, decimals=4, out=v)
    elif transform == "sqrt":
        with np.errstate(invalid="ignore"):
            v = np.sqrt(v)
    elif transform is not None:
        raise ValueError("unknown transform")
    return v


def bedgraph_bytes(chrom, starts, ends, values, transform=None) -> bytes:
    import pandas as pd

    v = transform_values(values, transform)
    df = pd.DataFrame({"Chromosome": [chrom] * len(v), "Start": np.asarray(starts, np.int64),
                       "End": np.asarray(ends, np.int64), "x": v})
    buf = io.StringIO()
    df[["Chromosome", "Start", "End", "x"]].to_csv(buf, sep="\t", header=False, index=False, float_format="%.4f",
                                                   lineterminator="\n")
    return buf.getvalue().encode("ascii")


def bedgraph_bytes_python(chrom, starts, ends, values, transform=None) -> bytes:
    v = transform_values(values, transform)
    rows = []
    for s, e, x in zip(np.asarray(starts, np.int64).tolist(), np.asarray(ends, np.int64).tolist(), v.tolist()):
        txt = "" if x != x else ("%.4f" % x)
        rows.append(f"{chrom}\t{s}\t{e}\t{txt}\n")
    return "".join(rows).encode("ascii")
