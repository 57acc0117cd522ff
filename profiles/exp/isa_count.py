#!/usr/bin/env python3
"""Instruction mix of the CMUX loop of k_blind_rotate<L,FAST> from the compiler's assembly.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o x.s tfhe_hip.hip
    python3 profiles/exp/isa_count.py x.s [L]

Blocks of the outer loop are weighted by their trip count per CMUX step (the inner digit-row loops run L-1 times).
"""
import collections
import re
import sys

path = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
name = f"_ZN4tfhe14k_blind_rotateILi{L}ELb1EEEvNS_15BlindRotateArgsE"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]

# basic blocks
blocks, cur = [], None
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        cur = {"label": m.group(1), "note": m.group(2) or "", "ins": []}
        blocks.append(cur)
    elif cur is not None and l.startswith("\t") and not l.startswith("\t."):
        op = l.strip().split()[0]
        if not op.startswith(";"):
            cur["ins"].append(op)

def klass(op):
    if op.startswith("v_") and ("f64" in op):
        if "fma" in op: return "v_fma_f64"
        if "mul" in op: return "v_mul_f64"
        if "add" in op: return "v_add_f64"
        return "v_other_f64"
    if op.startswith("v_"): return "v_int/other"
    if op.startswith("ds_"): return op.split("_b")[0] + ("_b128" if "b128" in op else "_b32" if "b32" in op else "")
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_"): return op
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "s_other"
    return op

# the outer loop = the depth-1 loop that contains nested loops (largest one)
loopblocks = [b for b in blocks if "Loop" in b["note"]]
tot = collections.Counter()
print(f"{'block':12s} {'weight':>6s} {'insts':>6s}  note")
for b in loopblocks:
    if len(b["ins"]) < 100:  # prologue helper loops
        continue
    inner = "Parent Loop" in b["note"]
    w = (L - 1) if inner else 1
    print(f"{b['label']:12s} {w:6d} {len(b['ins']):6d}  {b['note'][:60]}")
    for op in b["ins"]:
        tot[klass(op)] += w
valu = sum(v for k, v in tot.items() if k.startswith("v_"))
print(f"\nper CMUX step (L={L}): VALU {valu}, all {sum(tot.values())}")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v:6d}")
