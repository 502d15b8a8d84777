#!/usr/bin/env python3
"""Dynamic instruction counts along ONE path through a kernel's gfx950 assembly listing.
The listing is followed from a start label / line; at every conditional branch the next entry of a decision string says
whether it is taken ('t') or falls through ('n'); unconditional branches are followed.  Counts are kept per WV_MARK
region and per class (vector cheap / vector other / scalar / nop / wait / branch / LDS / vector memory).  When the
decisions run out, the tool prints the branch it stopped at with some context, so that a path can be worked out
interactively with the source next to it.
usage: isa_path.py build/enc4.s <start line> <stop regex> <decisions> [-v]
Decision strings worked out in rounds 3 and 4 (they hold while the branch structure of the kernel does -- the last cuts of
round 4 (run-length rows by quads, passes of noise, the mini-LZ's direct attempt, the decoders' plane forms) came after them, so
they have to be worked out again before they are quoted; the start line is the line of the named mark inside the kernel):
  tools/isa_path_rand12.decisions         encode_superblocks<4>, kernels.hip built as in csrc/Makefile: one pass of two int32
                                          blocks u & 0xFFF, from MARK load_block to MARK block_end (486 vector + 244 scalar)
  tools/isa_path_decode_rand12.decisions  decode_superblocks<4>, decode_kernels.hip built as in csrc/Makefile: one such
                                          block, from MARK dec_block_begin to the next one (82 vector + 75 scalar + 27 branches)"""
import re
import sys

src, start, stop_re, decisions = sys.argv[1], int(sys.argv[2]), re.compile(sys.argv[3]), sys.argv[4]
verbose = "-v" in sys.argv
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_cndmask_b32", "v_add_u16", "v_sub_u16"}
lines = open(src).read().split("\n")
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
regions = []
cur = ["(start)", {}]


def bump(k, n=1):
    cur[1][k] = cur[1].get(k, 0) + n


pc, di, steps = start - 1, 0, 0
while pc < len(lines):
    steps += 1
    if steps > 200000:
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
        print(f"{pc + 1:6d} {t}")
    if op == "s_branch":
        bump("branch")
        pc = labels[t.split()[1]]
        continue
    if op.startswith("s_cbranch"):
        bump("branch")
        if di >= len(decisions):
            print(f"--- undecided branch at line {pc + 1} (decision #{di}):")
            for k in range(max(0, pc - 14), min(len(lines), pc + 3)):
                print(f"{k + 1:6d} {lines[k]}")
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
    if op.startswith("v_"):
        bump("valu")
        cheap = base in CHEAP and not op.endswith(("_e64", "_sdwa", "_dpp"))
        bump("vcheap" if cheap else "vother")
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            bump("lane")
    elif op == "s_nop":
        bump("nop")
    elif op.startswith("s_waitcnt"):
        bump("wait")
    elif op.startswith("s_"):
        bump("salu")
    elif op.startswith("ds_"):
        bump("lds")
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        bump("vmem")
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
