#!/usr/bin/env python3
"""Instruction mix of the CMUX loop of k_blind_rotate<L, FAST>, read off the compiler's gfx950 assembly, for the
instantiation the reference's parameter sets of that l run: FAST (one-add rounding) at l = 3 (bgbit 6), the general
rounding at l = 2 (SECURITY_UINT1, bgbit 10) and l = 1 (bgbit 18 .. 23) -- tfhe_hip_ctx_create's fast_round rule.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o tfhe_hip.s tfhe_hip.hip
    python3 isa_mix.py tfhe_hip.s [--json kernel_isa.json]

`make` runs this after building the library, so bench.py prices the kernel against the FP64 vector roofline with
the instruction counts of the build it is timing (cross-check: SQ_INSTS_VALU per launch / (batch * n), profiles/).
Basic blocks of the CMUX loop are weighted by their trip count per step (the digit-row loops run l - 1 times).
"""
import collections
import json
import re
import sys


def klass(op):
    if op.startswith("v_") and "f64" in op:
        if "fma" in op:
            return "f64_fma"
        if "mul" in op:
            return "f64_mul"
        if "add" in op:
            return "f64_add"
        return "f64_other"  # conversions, rndne
    if op.startswith("v_"):
        return "valu_int"
    if op.startswith("ds_add") or op.startswith("ds_sub"):
        return "lds_atomic"
    if op.startswith("ds_"):
        wide = "b128" in op
        return ("lds_write" if "write" in op else "lds_read") + ("_b128" if wide else "_narrow")
    if op.startswith(("buffer_load", "global_load")):
        return "vmem_load"
    if op.startswith(("buffer_store", "global_store", "scratch_")):
        return "vmem_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def mix(lines, L):
    name = f"_ZN4tfhe14k_blind_rotateILi{L}ELb{1 if L == 3 else 0}EEEvNS_15BlindRotateArgsE"
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], None
    for l in lines[start:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = {"label": m.group(1), "note": m.group(2) or "", "ins": []}
            blocks.append(cur)
        elif cur is not None and l.startswith("\t") and not l.startswith("\t."):
            op = l.strip().split()[0]
            if not op.startswith(";"):
                cur["ins"].append(op)
    tot = collections.Counter()
    used = []
    for b in blocks:
        if "Loop" not in b["note"] or len(b["ins"]) < 100:  # the CMUX loop's blocks are hundreds of instructions each
            continue
        w = (L - 1) if "Parent Loop" in b["note"] else 1
        used.append((b["label"], w, len(b["ins"])))
        for op in b["ins"]:
            tot[klass(op)] += w
    out = dict(tot)
    out["valu"] = sum(v for k, v in tot.items() if k.startswith(("f64_", "valu_")))
    out["f64_flop_per_lane"] = 2 * tot["f64_fma"] + tot["f64_add"] + tot["f64_mul"]
    out["all"] = sum(tot.values())
    out["blocks"] = used
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    res = {f"l{L}": mix(lines, L) for L in (1, 2, 3)}
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    for k, r in res.items():
        print(k, {a: b for a, b in r.items() if a != "blocks"})


if __name__ == "__main__":
    main()
