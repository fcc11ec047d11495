#!/usr/bin/env python
"""Static instruction mix of a kernel between consecutive s_barrier instructions (from a -save-temps .s file).
usage: tools/isa_phases.py FILE.s KERNEL_SYMBOL_SUBSTRING"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].strip().endswith(":"))
rows, cur = [], collections.Counter()
loops = 0
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        rows.append(cur); cur = collections.Counter(); continue
    if op.startswith("v_mfma"): cur["mfma"] += 1
    elif op.startswith("v_accvgpr"): cur["acc_" + op.split("_")[2]] += 1; cur["valu"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): cur["vmem"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cur["br"] += 1
    elif op.startswith("s_nop"): cur["nop"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
rows.append(cur)
keys = ["mfma", "valu", "acc_read", "acc_write", "acc_mov", "ds", "vmem", "salu", "wait", "br", "nop"]
print("intv " + " ".join(f"{k:>9}" for k in keys))
tot = collections.Counter()
for i, r in enumerate(rows):
    print(f"{i:4d} " + " ".join(f"{r[k]:9d}" for k in keys)); tot.update(r)
print(" sum " + " ".join(f"{tot[k]:9d}" for k in keys))
