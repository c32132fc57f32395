#!/usr/bin/env python3
"""Generates tests/golden/oracle_v1.npz: small fixed cases (inputs are re-derived from seeds, outputs stored).

The reference (C#/.NET 9) cannot run in the build image and ships no golden vectors, so these vectors are outputs of
the CPU oracle (oracle/ga_oracle.cpp), which is itself pinned against numpy/scipy analytic models in
tests/test_oracle_*.py.  They freeze the oracle's behaviour (regression pin, checked bit-exactly on CPU) and give the
GPU tests a reference that needs no oracle build.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import _cases  # noqa: E402
from tests._oracle import OracleContext  # noqa: E402


def main():
    out = {}
    for name, (builder, frames) in _cases.CASES.items():
        ctx = OracleContext(48000)
        ch = builder(ctx)
        buf = np.zeros((ch, frames), np.float32)
        ctx.Render(buf, frames)
        ctx.Dispose()
        out[name] = buf
        print(f"{name:28s} ch={ch} frames={frames} rms={np.sqrt(np.mean(buf.astype(np.float64) ** 2)):.4e}")
    path = os.path.join(ROOT, "tests", "golden", "oracle_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
