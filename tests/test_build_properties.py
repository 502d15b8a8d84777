"""Properties of the gfx950 build that the measured numbers rest on, checked with the cross compiler (no GPU).
decode_kernels.hip is compiled with -mllvm -structurizecfg-skip-uniform-regions (csrc/Makefile): that only pays while
decode_superblocks has no divergent branch at all, which LLVM's uniformity analysis can tell."""
import os
import shutil
import subprocess

import pytest

from _libs import ROOT

OPT = "/opt/rocm/lib/llvm/bin/opt"


@pytest.mark.skipif(shutil.which("hipcc") is None or not os.path.exists(OPT), reason="needs hipcc and opt of the ROCm toolchain")
def test_decode_kernels_have_no_divergent_branch():
    keys = ["decode_superblocksILj2E", "decode_superblocksILj4E", "decode_superblocksILj8E", "decode_superblocksILj0E"]
    p = subprocess.run([os.path.join(ROOT, "tools", "divergent_branches.sh"), "decode_kernels.hip"] + keys, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-800:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert [l for l in lines if l.startswith("== ")] == ["== " + k for k in keys], p.stdout[-800:]
    assert [l for l in lines if not l.startswith("== ")] == [], "divergent branches:\n" + p.stdout[-1500:]
