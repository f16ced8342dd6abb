"""-m gpu: a fixed-seed slice of the three parity fuzzers (tools/fuzz_parity.py, fuzz_long.py, fuzz_exact.py), so that random
shapes / hops / row counts / segment lengths / signal kinds are part of the driver-run evidence and not only of the
builder's logs.  Each fuzzer draws its cases from numpy's Generator(seed), runs them through the C ABI
on the MI355X and compares with the oracle's bit models: float32 mode - (column,row) and per-bin power array_equal, dB within
8.7e-4, palette index within +-1; EXACT mode - every output array_equal (dump, dB bits, palette index, streaming == batch).
The fuzzers print one line per failing case and exit 1; a whole slice takes ~10 s."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, cases, seed, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("EMSPEC_SEGLEN", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert f"{cases} cases, 0 failures" in r.stdout, tail
    return r.stdout


@pytest.mark.timeout(900)
def test_fuzz_slice_float32_shapes():
    """150 random (N, hop, rows, reassign, streams, frames, signal kind, display settings) cases: batch columns, the per-bin
    dump of the last frames, and every fourth case the streaming call against the batch."""
    _run("fuzz_parity.py", 150, 505)


@pytest.mark.timeout(900)
def test_fuzz_slice_fused_long_walks():
    """40 random long walks (hundreds to thousands of frames, several segments per stream, forced segment lengths through the
    diagnostic build) over every fused float32 shape incl. N = 16384."""
    _run("fuzz_long.py", 40, 506)


@pytest.mark.timeout(900)
def test_fuzz_slice_exact_mode():
    """150 random EXACT-mode cases, byte-equal to the binary64 bit model; (column,row) also against the independent float64
    three-window method (0 mismatches expected: asserted from the tool's summary line)."""
    out = _run("fuzz_exact.py", 150, 507, {"EMSPEC_FUZZ_DIAG": "1"})
    assert "vs the independent float64 method: 0 mismatches" in out, out[-1500:]


@pytest.mark.timeout(900)
def test_fuzz_slice_live_multi_stream():
    """60 random live sessions (round 6: emspec_columns / emspec_push_samples_multi / flush / reset_stream; either mode, any
    size, 1..8 streams, sub-hop and over-long blocks, page-locked or pageable buffers, one stream restarted mid-session) against
    the oracle's batch columns: EXACT bytes, float32 within 8.7e-4 dB."""
    _run("fuzz_live.py", 60, 508)
