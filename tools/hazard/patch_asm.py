#!/usr/bin/env python3
"""Patch the gfx950 assembly of the policy translation unit (tools/hazard/build_variant.sh).  Specs:
  after:<class>:<n>   insert `s_nop n` AFTER every instruction of the class
  before:<class>:<n>  ... BEFORE it
classes: pk_opsel = packed fp32 / pk_mov with a cross-register op_sel (the SLP vectoriser's), pk = every v_pk_*_f32 / v_pk_mov_b32,
         mfma = every v_mfma_*, mfma_last = an MFMA that is not directly followed by another MFMA (the tail of a chain)
Only the matrix-core rollout kernels (k_rollout_pair*, k_rollout<*, 4|8>) matter; every kernel of the file is patched."""
import re
import sys

path, specs = sys.argv[1], sys.argv[2:]
lines = open(path).read().split("\n")
CLS = {
    "pk_opsel": re.compile(r"^\s+v_pk_(fma|mul|add)_f32 .*op_sel:\[|^\s+v_pk_mov_b32 "),
    "pk": re.compile(r"^\s+v_pk_(fma|mul|add)_f32 |^\s+v_pk_mov_b32 "),
    "mfma": re.compile(r"^\s+v_mfma_"),
}
is_instr = re.compile(r"^\s+[vsdb][a-z_]")
for spec in specs:
    where, cls, n = spec.split(":")
    n = int(n)
    out, count = [], 0
    for i, l in enumerate(lines):
        hit = False
        if cls == "mfma_last":
            if CLS["mfma"].match(l):
                j = i + 1
                while j < len(lines) and not is_instr.match(lines[j]):
                    j += 1
                hit = j >= len(lines) or not CLS["mfma"].match(lines[j])
        else:
            hit = bool(CLS[cls].match(l))
        if hit and where == "before":
            out.append(f"\ts_nop {n}")
        out.append(l)
        if hit and where == "after":
            out.append(f"\ts_nop {n}")
        count += hit
    lines = out
    print(f"{spec}: {count} sites")
open(path, "w").write("\n".join(lines))
