#!/usr/bin/env python3
"""Instruction mix of a line range of hipcc -S output (tools/isa_mix.py file.s first last): counts by issue class with the
per-class issue cost on a gfx950 SIMD-32 (plain VALU 2 cycles per wave64 instruction, packed-f32 / DPP 4, transcendental 4)."""
import re, sys, collections
def classify(op, line):
    if op.startswith("v_pk_"): return "valu_pk"
    if "dpp" in line or op.endswith("_dpp"): return "valu_dpp"
    if op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith("v_"): return "valu"
    if op.startswith(("buffer_load", "global_load", "flat_load")): return "vmem_rd"
    if op.startswith(("buffer_store", "global_store", "flat_store")): return "vmem_wr"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"
COST = {"valu": 2, "valu_pk": 4, "valu_dpp": 4, "valu_trans": 4}
def mix(lines):
    c = collections.Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"): continue
        op = l.split()[0]
        c[classify(op, l)] += 1
    return c
if __name__ == "__main__":
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    L = open(f).read().split("\n")[a - 1:b]
    c = mix(L)
    tot = sum(COST.get(k, 0) * v for k, v in c.items())
    print(dict(c), "VALU issue cycles:", tot)
