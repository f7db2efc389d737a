#!/usr/bin/env python3
"""Instruction-class pattern of the basic blocks of one kernel in a hipcc -S dump.
usage: tools/isa_pattern.py file.s mangled-name-substring [min-mfma]
M mfma, d ds_read, D ds_write, G global_load_lds, g global/buffer load, W store, w s_waitcnt, a accvgpr move,
n s_nop, v other VALU, s SALU, b branch"""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r"^(\S*%s\S*):" % re.escape(sub), s, re.M)
i = m.start(); j = s.index(".Lfunc_end", i)
blocks = []; cur = [m.group(1)]
for l in s[i:j].splitlines()[1:]:
    if re.match(r"^\.LBB", l):
        blocks.append(cur); cur = [l]
    else:
        cur.append(l)
blocks.append(cur)
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("ds_read"): return "d"
    if op.startswith("ds_"): return "D"
    if op.startswith("global_load_lds"): return "G"
    if op.startswith(("global_load", "buffer_load")): return "g"
    if op.startswith(("global_store", "buffer_store")): return "W"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("s_nop"): return "n"
    if op.startswith(("s_cbranch", "s_branch")): return "b"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "s"
    return ""
for b in blocks:
    seq = "".join(cls(l.split()[0]) for l in b[1:] if l.strip() and not l.strip().startswith((";", ".")))
    if seq.count("M") >= minm:
        print(b[0], len(seq), {c: seq.count(c) for c in sorted(set(seq))})
        for k in range(0, len(seq), 128):
            print("   ", seq[k:k + 128])
