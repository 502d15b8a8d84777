#!/usr/bin/env python3
"""Static instruction counts between the WV_MARK comments of one kernel in a gfx950 assembly listing.
usage: isa_regions.py build/kernels.s encode_superblocksILj4E
Each region runs from its mark to the next mark in program order (code the compiler moved across a mark is
attributed to where it ended up); inner branches are not followed, so counts are upper bounds of one pass."""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_cndmask_b32", "v_add_u16", "v_sub_u16"}
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # (a kernel may hold several s_endpgm)
regions, cur = [], ["(prologue)", {}]
def bump(d, k, n=1):
    d[k] = d.get(k, 0) + n
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r";+\s*MARK (\S+)", t)
    if m:
        regions.append(cur)
        cur = [m.group(1), {}]
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    d = cur[1]
    if op.startswith("v_"):
        bump(d, "valu")
        cheap = base in CHEAP and not op.endswith(("_e64", "_sdwa", "_dpp"))
        bump(d, "cyc", 2.3 if cheap else 4.2)
    elif op in ("s_nop",):
        bump(d, "nop")
    elif op.startswith("s_waitcnt"):
        bump(d, "wait")
    elif op.startswith("s_cbranch") or op == "s_branch":
        bump(d, "branch")
    elif op.startswith("s_"):
        bump(d, "salu")
    elif op.startswith("ds_"):
        bump(d, "lds")
    elif op.startswith(("global_", "buffer_", "flat_")):
        bump(d, "vmem")
regions.append(cur)
print(f"{'region':18s} {'valu':>5s} {'~cyc':>6s} {'salu':>5s} {'nop':>4s} {'wait':>4s} {'br':>4s} {'lds':>4s} {'vmem':>4s}")
for name, d in regions:
    print(f"{name:18s} {d.get('valu',0):5d} {d.get('cyc',0):6.0f} {d.get('salu',0):5d} {d.get('nop',0):4d} {d.get('wait',0):4d} {d.get('branch',0):4d} {d.get('lds',0):4d} {d.get('vmem',0):4d}")
