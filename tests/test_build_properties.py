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


def _device_asm(source, flags):
    out = os.path.join(ROOT, "build", "props_" + os.path.basename(source) + ".s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", os.path.join(ROOT, "stenos_amd", "csrc", source), "-o", out] + flags
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    with open(out) as f:
        text = f.read()
    os.remove(out)  # (build/ travels to the GPU box: no listings left behind)
    return text.split("\n")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("source,flags", [("kernels.hip", ["-DWV_PREDICATE_BRANCHES"]), ("decode_kernels.hip", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"])])
def test_wide_stores_in_asm_statements_keep_their_wait_states(source, flags):
    """On gfx940 and later a vector-memory store of more than 8 bytes still reads its data registers for two cycles after
    it issues; an instruction that writes them sooner corrupts the store.  The compiler inserts the wait states for its own
    stores, but it does not look into asm statements (wavevec.h: predicated, streamed and write-through stores), so those
    must carry them: behind every 12- or 16-byte store inside an asm statement there have to be two wait states (s_nop 1,
    or two scalar instructions) before the statement ends.  Found the hard way: 16-byte write-through stores of the
    speculative copy followed by a v_mov into their first data register."""
    lines = _device_asm(source, flags)
    inside, stores, bad = False, 0, []
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inside, pending = True, None
            continue
        if t.startswith(";;#ASMEND"):
            if inside and pending is not None and pending > 0:
                bad.append((i + 1, lines[max(0, i - 4):i + 1]))
            inside = False
            continue
        if not inside or not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op in ("global_store_dwordx4", "global_store_dwordx3", "flat_store_dwordx4", "flat_store_dwordx3"):
            stores += 1
            pending = 2
        elif pending is not None and pending > 0:
            if op == "s_nop":
                pending -= int(t.split()[1]) + 1
            elif op.startswith("s_"):
                pending -= 1
            else:
                pending = 99  # a vector instruction right behind the store
    assert stores > 0, "no wide store found inside asm statements: has the check lost its target?"
    assert not bad, bad[:3]


@pytest.mark.skipif(shutil.which("python3") is None, reason="needs python3")
def test_shape_tables_are_current():
    """stenos_amd/csrc/shape_tables.inc is what tools/gen_shape_tables.py writes (slot_codec.h reads it per pass)."""
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_shape_tables.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, "run tools/gen_shape_tables.py"
