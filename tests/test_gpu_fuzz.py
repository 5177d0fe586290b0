"""Randomised parity stress on the GPU: scripts/fuzz_gemm.py (random shapes, layouts, element offsets,
leading-dimension padding, alpha/beta, tile configurations) must report zero mismatches against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_gemm_matches_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_gemm.py"), "120", str(seed)],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and "0 failures" in r.stdout, tail


@pytest.mark.gpu
def test_fuzz_conv_matches_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_conv.py"), "80", "5"],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and "0 failures" in r.stdout, tail
