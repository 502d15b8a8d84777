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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("source,flags", [("kernels.hip", ["-DWV_PREDICATE_BRANCHES"]), ("decode_kernels.hip", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"])])
def test_vector_memory_in_asm_statements_waits_for_its_scalar_base(source, flags):
    """A vector-memory instruction may read a scalar register five wait states after a VALU instruction -- a v_readfirstlane
    that made a pointer uniform -- wrote it, at the earliest (gfx9 family), and the compiler does not look into asm statements:
    one that addresses through a scalar base carries the wait states itself (wavevec.h: s_nop 4 in front of the
    write-through stores, s_nop 2 behind the two scalar instructions of the streamed ones).  Found the hard way: a memory
    fault at address 0 from the speculative copy once the scheduler put the v_readfirstlane right in front of the store."""
    import re

    lines = _device_asm(source, flags)
    inside, found, bad = False, 0, []
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inside, waited = True, 0
            continue
        if t.startswith(";;#ASMEND"):
            inside = False
            continue
        if not inside or not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op.startswith(("global_store", "global_load", "global_atomic")) and re.search(r"\bs\[\d+:\d+\]", t):
            found += 1
            if waited < 5:
                bad.append((i + 1, t))
        elif op == "s_nop":
            waited += int(t.split()[1]) + 1
        elif op.startswith("s_"):
            waited += 1
        else:
            waited = 0  # (a vector instruction inside the statement: it could be the writer)
    assert found > 0, "no scalar-based vector-memory instruction found inside asm statements: has the check lost its target?"
    assert not bad, bad[:3]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("source,flags", [("kernels.hip", ["-DWV_PREDICATE_BRANCHES"]), ("decode_kernels.hip", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"])])
def test_asm_statements_that_change_the_exec_mask_put_it_back(source, flags):
    """Predicated accesses and the chain walks of the decoders (wavevec.h: lds_rle_walk16, lds_lz_walk32) narrow the exec mask
    inside an asm statement; the compiler does not know, so every such statement has to save the mask first and to end with
    the instruction that restores it from the same registers."""
    lines = _device_asm(source, flags)
    inside, body, seen, walks = False, [], 0, 0
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inside, body = True, []
            continue
        if t.startswith(";;#ASMEND"):
            inside = False
            writes = [b for b in body if b.replace(",", " ").split()[1:2] == ["exec"] and b.split()[0].startswith("s_")]
            if writes:
                seen += 1
                walks += any(b.startswith("s_lshl_b64 exec") for b in body)
                saves = [b for b in body if b.startswith("s_mov_b64") and b.replace(",", " ").split()[2:3] == ["exec"]]
                assert saves and body.index(saves[0]) < body.index(writes[0]), (i + 1, body[:6])
                saved = saves[0].replace(",", " ").split()[1]
                assert body[-1].replace(",", " ").split() == ["s_mov_b64", "exec", saved], (i + 1, body[-3:])
            continue
        if inside and t and not t.startswith(";"):
            body.append(t)
    assert seen > 0
    if source == "decode_kernels.hip":
        assert walks >= 2  # (both walks are in the build)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_kernel_resources_stay_inside_their_budget():
    """What the measured numbers rest on (-Rpass-analysis=kernel-resource-usage): the decoders and the hot encoders keep
    their occupancy, the fused encoders of bytesoftype 2 and 8 and every decoder use no scratch memory, and the scratch of the
    int32 encoder stays what the cold mini-LZ branches spill (its common pass executes none of it: tools/isa_path2.py), within
    a stated budget -- a change that pushes the hot loop into scratch shows up here, not three rounds later in a profile."""
    import re

    def usage(source, flags):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-c", os.path.join(ROOT, "stenos_amd", "csrc", source),
               "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + flags
        err = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
        res, cur = {}, None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = res.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        return res

    enc = usage("kernels.hip", ["-DWV_PREDICATE_BRANCHES"])
    dec = usage("decode_kernels.hip", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"])

    def one(res, key):
        hits = [v for k, v in res.items() if key in k]
        assert len(hits) == 1, (key, [k for k in res if key in k])
        return hits[0]

    for T in (2, 4, 8, 0):
        d = one(dec, f"decode_superblocksILj{T}E")
        assert d["ScratchSize"] == 0 and d["VGPRs Spill"] == 0, (T, d)
    for T, occ in ((2, 8), (4, 8), (8, 5)):
        e = one(enc, f"encode_superblocksILj{T}E")
        assert e["Occupancy"] == occ, (T, e)
        assert e["TotalSGPRs"] <= 80 or T == 8, (T, e)  # (above 80 a CU admits seven 256-thread workgroups, not eight)
    e2 = one(enc, "encode_superblocksILj2E")
    assert e2["ScratchSize"] <= 8 and e2["VGPRs Spill"] <= 2, e2  # (one register parked in the prologue and read back once, behind a pass's emission)
    assert one(enc, "encode_superblocksILj8E")["ScratchSize"] == 0
    e4 = one(enc, "encode_superblocksILj4E")
    assert e4["ScratchSize"] <= 32 and e4["VGPRs Spill"] <= 10, e4


def test_no_experiment_switch_in_the_product_sources():
    """Timing experiments (builds that write wrong frames on purpose) live in git history and DESIGN.md, not behind -D switches
    in the sources the product is built from: a stray define must not be able to ship a broken codec."""
    import glob
    import re

    for path in glob.glob(os.path.join(ROOT, "stenos_amd", "csrc", "*")) + [os.path.join(ROOT, "stenos_amd", "csrc", "Makefile")]:
        if not os.path.isfile(path) or path.endswith(".inc"):
            continue
        text = open(path).read()
        assert not re.search(r"STENOS_(EXP|PAD)_\w+", text), path


@pytest.mark.skipif(shutil.which("python3") is None, reason="needs python3")
def test_shape_tables_are_current():
    """stenos_amd/csrc/shape_tables.inc is what tools/gen_shape_tables.py writes (slot_codec.h reads it per pass)."""
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_shape_tables.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, "run tools/gen_shape_tables.py"
