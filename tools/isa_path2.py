#!/usr/bin/env python3
"""isa_path.py with source attribution: the listing comes from a build with -gline-tables-only, so every instruction
carries the `.loc` (file, line) in force; dynamic counts along the chosen path are summed per MARK region, per source
line and per source function-ish bucket (file:line ranges given on the command line are not needed: the per-line table
is printed sorted).  At an undecided branch the tool prints the context WITH the source lines of the instructions, which
is what makes the decisions quick to work out.
usage: isa_path2.py encg.s <start line> <stop regex> <decisions> [-v] [-lines N]
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWV_PREDICATE_BRANCHES -DWV_MARKS --cuda-device-only
       -gline-tables-only -S stenos_amd/csrc/kernels.hip -o encg.s"""
import re
import sys

src, start, stop_re, decisions = sys.argv[1], int(sys.argv[2]), re.compile(sys.argv[3]), sys.argv[4]
verbose = "-v" in sys.argv
nlines = int(sys.argv[sys.argv.index("-lines") + 1]) if "-lines" in sys.argv else 40
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_cndmask_b32", "v_add_u16", "v_sub_u16",
         "v_bitop3_b32", "v_pk_min_u16", "v_pk_max_u16", "v_pk_sub_u16", "v_ashrrev_i32", "v_lshlrev_b16", "v_lshrrev_b16"}
lines = open(src).read().split("\n")
files = {}
labels = {}
loc_at = [None] * len(lines)  # (file, line) in force at each listing line
cur_loc = None
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur_loc = (int(m.group(1)), int(m.group(2)))
    loc_at[i] = cur_loc


def locname(i):
    c = loc_at[i]
    return f"{files.get(c[0], c[0])}:{c[1]}" if c else "?"


regions = []
cur = ["(start)", {}]
perline = {}


def bump(k, n=1):
    cur[1][k] = cur[1].get(k, 0) + n


pc, di, steps = start - 1, 0, 0
while pc < len(lines):
    steps += 1
    if steps > 400000:
        print("too many steps")
        break
    l = lines[pc]
    t = l.strip()
    m = re.match(r";+\s*MARK (\S+)", t)
    if m:
        regions.append(cur)
        cur = [m.group(1), {}]
        if stop_re.search(m.group(1)) and steps > 5:
            break
        pc += 1
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        pc += 1
        continue
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if verbose:
        print(f"{pc + 1:6d} {locname(pc):28s} {t}")
    if op == "s_branch":
        bump("branch")
        pc = labels[t.split()[1]]
        continue
    if op.startswith("s_cbranch"):
        bump("branch")
        if di >= len(decisions):
            print(f"--- undecided branch at line {pc + 1} (decision #{di}):")
            for k in range(max(0, pc - 22), min(len(lines), pc + 2)):
                s = lines[k].strip()
                if s and not s.startswith(".") :
                    print(f"{k + 1:6d} {locname(k):28s} {lines[k]}")
            tgt = labels[t.split()[1]]
            print(f"   taken -> line {tgt + 1} ({locname(tgt + 1)}); fall through -> {locname(pc + 1)}")
            break
        d = decisions[di]
        di += 1
        if d == "t":
            pc = labels[t.split()[1]]
        else:
            pc += 1
        continue
    if op == "s_endpgm":
        break
    kind = None
    if op.startswith("v_"):
        bump("valu")
        cheap = base in CHEAP and not op.endswith(("_e64", "_sdwa", "_dpp"))
        bump("vcheap" if cheap else "vother")
        kind = "vc" if cheap else "vo"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            bump("lane")
    elif op == "s_nop":
        bump("nop")
        kind = "s"
    elif op.startswith("s_waitcnt"):
        bump("wait")
        kind = "s"
    elif op.startswith("s_"):
        bump("salu")
        kind = "s"
    elif op.startswith("ds_"):
        bump("lds")
        kind = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        bump("vmem")
        kind = "vm"
    if kind:
        d = perline.setdefault(locname(pc), {})
        d[kind] = d.get(kind, 0) + 1
    pc += 1
regions.append(cur)
tot = {}
print(f"{'region':18s} {'valu':>5s} {'cheap':>5s} {'other':>5s} {'lane':>4s} {'salu':>5s} {'nop':>4s} {'wait':>4s} {'br':>4s} {'lds':>4s} {'vmem':>4s}")
for name, d in regions:
    if not d:
        continue
    print(f"{name:18s} {d.get('valu',0):5d} {d.get('vcheap',0):5d} {d.get('vother',0):5d} {d.get('lane',0):4d} {d.get('salu',0):5d} {d.get('nop',0):4d} {d.get('wait',0):4d} {d.get('branch',0):4d} {d.get('lds',0):4d} {d.get('vmem',0):4d}")
    for k, v in d.items():
        tot[k] = tot.get(k, 0) + v
print(f"{'TOTAL':18s} {tot.get('valu',0):5d} {tot.get('vcheap',0):5d} {tot.get('vother',0):5d} {tot.get('lane',0):4d} {tot.get('salu',0):5d} {tot.get('nop',0):4d} {tot.get('wait',0):4d} {tot.get('branch',0):4d} {tot.get('lds',0):4d} {tot.get('vmem',0):4d}")
print("decisions used:", di, "of", len(decisions), " stopped at line", pc + 1)
print(f"\nper source line (vector cheap / vector other / scalar / lds / vmem), top {nlines}:")
rows = sorted(perline.items(), key=lambda kv: -(kv[1].get("vc", 0) + kv[1].get("vo", 0) + kv[1].get("s", 0)))
for name, d in rows[:nlines]:
    print(f"  {name:28s} {d.get('vc',0):4d} {d.get('vo',0):4d} {d.get('s',0):4d} {d.get('lds',0):3d} {d.get('vm',0):3d}")
