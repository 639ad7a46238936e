"""Race screen for the kernels that order their LDS traffic by hand (LDS-DMA with counted vmcnt + raw s_barrier: conv_igemm
`dma` / `dmai` in the f16 and i8 modes, conv1x1_areg.hip, conv1x1_b2b.hip, conv1x1_q8.hip, conv_hl.hip of the f16hl mode): the same frame many times on three contexts running at the same time -- every run must
give the same bits.  A read placed one phase too early passes single runs whenever the DMA happens to land first
(cdna_hip_programming.md); it shows up as a second distinct result under varying timing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_repeated_concurrent_runs_are_bit_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "race_screen.py"), "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RACE SCREEN CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
