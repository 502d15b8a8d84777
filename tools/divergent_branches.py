#!/usr/bin/env python3
"""Helper of tools/divergent_branches.sh: the DIVERGENT terminators of one function in the output of
opt -passes='print<uniformity>', each printed as function:line <- caller:line <- ... from the !dbg locations of the same
branch in the IR file.  usage: divergent_branches.py <file.ll> <uniformity.txt> <function name fragment>"""
import re
import sys

ll = open(sys.argv[1]).read()
key = sys.argv[3]
loc, scope = {}, {}
for m in re.finditer(r"^!(\d+) = (?:distinct )?!DILocation\(line: (\d+), column: \d+, scope: !(\d+)(?:, inlinedAt: !(\d+))?\)", ll, re.M):
    loc[m.group(1)] = (int(m.group(2)), m.group(3), m.group(4))
for m in re.finditer(r'^!(\d+) = distinct !DISubprogram\(name: "([^"]+)"', ll, re.M):
    scope[m.group(1)] = m.group(2)
for m in re.finditer(r"^!(\d+) = (?:distinct )?!DILexicalBlock(?:File)?\(scope: !(\d+)", ll, re.M):
    scope[m.group(1)] = ("->", m.group(2))


def function_of(s):
    while isinstance(scope.get(s), tuple):
        s = scope[s][1]
    return scope.get(s, "?")


def chain(n):
    out = []
    while n and n in loc:
        line, s, inlined = loc[n]
        out.append(f"{function_of(s)}:{line}")
        n = inlined
    return " <- ".join(out)


body = re.search(r"^define [^\n]*@[^\n(]*" + re.escape(key) + r"[^\n]*\{\n(.*?)^\}", ll, re.M | re.S).group(1)
branches = {}
for l in body.split("\n"):
    t = l.strip()
    if t.startswith("br i1"):
        d = re.search(r"!dbg !(\d+)", t)
        branches[t.split(", !dbg")[0].split(", !llvm.loop")[0]] = d.group(1) if d else None
inside = False
for line in open(sys.argv[2]):
    if line.startswith("UniformityInfo for function"):
        inside = key in line
    if inside and "DIVERGENT:" in line and " br i1" in line:
        t = line.split("DIVERGENT:")[1].strip().split(", !dbg")[0].split(", !llvm.loop")[0]
        n = branches.get(t)
        print(chain(n) if n else "(no line) " + t)
