#!/usr/bin/env python3
"""developer aid: which VGPRs are live across a stretch of k_solve's assembly without being used in it.
usage: live.py engine.s [kernel-substring]   -> finds the densest v_fma_f64 region (the LU's dense block) and reports."""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
kname = sys.argv[2] if len(sys.argv) > 2 else "k_solve"
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+%s\S*:" % kname, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
rx = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
def regs(l):
    out = set()
    l = l.split(";")[0]
    for m in rx.finditer(l):
        if m.group(1): out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
occ = [regs(l) for l in body]
first, last = {}, {}
for i, r in enumerate(occ):
    for v in r:
        first.setdefault(v, i); last[v] = i
# densest fma region: window of 400 lines with most v_fma_f64
fma = [1 if ("v_mul_f64" in l or "v_fma_f64" in l) else 0 for l in body]
W = 300
best, bi = -1, 0
s = sum(fma[:W])
for i in range(len(body) - W):
    if s > best: best, bi = s, i
    s += fma[i + W] - fma[i]
print("kernel lines", len(body), "densest fma window at", bi, "with", best, "fma")
used = set().union(*occ[bi:bi + W])
live = [v for v in first if first[v] < bi and last[v] > bi + W]
idle = sorted(v for v in live if v not in used)
print("VGPRs used in window:", len(used), " live across and unused in it:", len(idle))
# what are the idle ones? show their last definition before the window (dst = first operand)
defs = collections.Counter()
for v in idle:
    for i in range(bi, -1, -1):
        l = body[i].split(";")[0].strip()
        m = re.match(r"(\S+)\s+(v\d+|v\[\d+:\d+\])", l)
        if m and v in regs(m.group(2)):
            defs[m.group(1)] += 1
            break
print(defs.most_common(20))
# readlane/writelane spill traffic in the whole kernel
print("v_writelane", sum("v_writelane" in l for l in body), "v_readlane", sum("v_readlane" in l for l in body), "scratch_", sum("scratch_" in l for l in body))
if len(sys.argv) > 3:
    for v in idle:
        for i in range(bi, -1, -1):
            l = body[i].split(";")[0].strip()
            m = re.match(r"(\S+)\s+(v\d+|v\[\d+:\d+\])", l)
            if m and v in regs(m.group(2)):
                print("v%d @%d: %s   (last use @%d: %s)" % (v, i, l, last[v], body[last[v]].strip()[:70]))
                break
