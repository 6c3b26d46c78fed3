"""Compiles tiler_slider_amd/csrc/ts_core.h for the HOST (tests/native/core_check.cpp) and checks the
kernels' sort-free transition arithmetic — the one-register bitboard form (S <= 8) and the
line-mask form (S <= 32) — against the oracle's restatement of the reference mechanics on random
boards.  Runs in the GPU-less build container; it is a test harness, not a product path."""
import os
import subprocess

from conftest import ROOT


def test_closed_form_matches_reference_mechanics(tmp_path):
    exe = str(tmp_path / "core_check")
    native = os.path.join(ROOT, "tests", "native", "core_check.cpp")
    oracle_c = os.path.join(ROOT, "oracle", "ts_oracle.c")
    obj = str(tmp_path / "ts_oracle.o")
    subprocess.run(["gcc", "-O2", "-std=c11", "-c", oracle_c, "-o", obj], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", native, obj, "-o", exe], check=True)
    out = subprocess.run([exe, "6000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout
