"""Golden fixtures (tests/golden/oracle_v1.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them bit-exactly (regression pin of the restatement).
GPU: the HIP path must match them (bit-exact where the path is elementwise, <= 1e-5 RMS for the convolver) -- this
test does not need the oracle at run time.
"""
import os

import numpy as np
import pytest

from tests import _cases
from tests import _graphs as G

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_v1.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", list(_cases.CASES))
def test_oracle_reproduces_golden(golden, name):
    from tests._oracle import OracleContext
    builder, frames = _cases.CASES[name]
    ctx = OracleContext(48000)
    ch = builder(ctx)
    out = G.render(ctx, ch, frames)
    assert np.array_equal(out, golden[name])


EXACT = {"plumbing", "biquad", "scheduling", "source_replay"}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_cases.CASES))
def test_hip_matches_golden(golden, name):
    from graphaudio_amd import OfflineAudioContext
    builder, frames = _cases.CASES[name]
    ctx = OfflineAudioContext(48000)
    ch = builder(ctx)
    out = G.render(ctx, ch, frames)
    ref = golden[name]
    if name in EXACT:
        assert np.array_equal(out, ref)
    else:
        err = G.rms(out - ref)
        assert err <= 1e-5 and err <= 3e-6 * max(G.rms(ref), 1e-3), err
