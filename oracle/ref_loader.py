"""TEST INFRASTRUCTURE ONLY -- import the REAL reference hot-path module built by `make -C oracle ref`.

Only usable in the build container (needs /root/reference for the package's pure-Python siblings and
oracle/_ref for the compiled extensions).  Returns None when unavailable; callers must then fall back to
the committed golden vectors.  Nothing on the GPU box may call this with an expectation of success.
"""
from __future__ import annotations

import glob
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
REF_SO_DIR = os.path.join(_HERE, "_ref")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "consenrich")) and bool(
        glob.glob(os.path.join(REF_SO_DIR, "cconsenrich*.so"))
    )


def load():
    """Return the reference's ``consenrich.cconsenrich`` module or None."""
    if not available():
        return None
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import consenrich  # the reference package's own lightweight __init__ (lazy exports only)

    if REF_SO_DIR not in consenrich.__path__:
        consenrich.__path__.append(REF_SO_DIR)
    from consenrich import cconsenrich  # noqa: PLC0415

    return cconsenrich
