"""Static checks on the gfx950 code the compiler emits for the product kernels (no GPU: hipcc cross-compiles)."""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_long_read_merge_compiles_to_returning_adds_in_front_of_the_counter():
    """k_read_stats merges the slices of a long read without a fence: the ordering rests on the per-bin adds being RETURNING
    agent-scope atomics that are waited for before the slices-done counter is added to (pg_kernels.hip, `if (m.split)`).
    Round 5's form was constant-folded into non-returning adds; the exact median (gmove.cpp:142-184) depends on it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/isa_stats.py"), "--check-long-merge"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "17 returning per-bin adds" in r.stdout
