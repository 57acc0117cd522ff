#!/usr/bin/env python3
"""One profiles/pmc_roofline.json entry from a tag's PMC summary:

    python3 profiles/pmc_roofline.py <tag>_pmc.csv <tag>_kernel_stats.csv <tag> [bench args]  > entry.json

Per launch of each of the two kernels of the path: HBM-side bytes = FETCH_SIZE x 2 + WRITE_SIZE (both reported in
KiB; the x2 is the gfx950 correction for 16-byte coalesced streams, MI355X_MICROARCH.md section HBM), the
instruction counters, and the fraction of the SIMDs' issue slots used (VALU + scalar + LDS instructions issued
per wave-cycle resident: SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES x waves per SIMD is what the SQ itself reports;
`issue_frac` below is (SQ_INSTS_VALU x 4) / SQ_BUSY_CYCLES-normalised SIMD cycles)."""
import argparse
import csv
import json
import sys

ap = argparse.ArgumentParser()
ap.add_argument("pmc")
ap.add_argument("stats")
ap.add_argument("tag")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--params", default="SECURITY_128_BIT")
ap.add_argument("--gate", default="nand")
args, _ = ap.parse_known_args()

rows = list(csv.DictReader(open(args.pmc)))
stats = list(csv.DictReader(open(args.stats)))


def per_launch(kernel_sub, counter):
    for r in rows:
        if kernel_sub in r["kernel"] and r["counter"] == counter:
            return float(r["per_launch"])
    return None


def avg_us(kernel_sub):
    for r in stats:
        if kernel_sub in r["kernel"]:
            return float(r["avg_us"])
    return None


def kernel_entry(sub):
    fetch, write = per_launch(sub, "FETCH_SIZE"), per_launch(sub, "WRITE_SIZE")
    e = {"kernel_match": sub, "avg_launch_us_rocprof": avg_us(sub)}
    if fetch is not None and write is not None:
        e["fetch_size_kib"], e["write_size_kib"] = fetch, write
        e["hbm_bytes_per_launch"] = int((2 * fetch + write) * 1024)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
              "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
              "SQ_WAIT_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
        v = per_launch(sub, c)
        if v is not None:
            e[c] = v
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                  "SQ_WAIT_ANY"):
            if c in e:
                e[c.lower() + "_per_wave_cycle"] = round(e[c] / wc, 4)
        # per-wave active fraction x resident waves per SIMD = share of the SIMD's issue cycles (SQ_ACTIVE_* count 4-cycle quads)
        if "SQ_ACTIVE_INST_ANY" in e:
            e["issue_frac"] = round(e["SQ_ACTIVE_INST_ANY"] / wc, 4)
    if "TCP_TCC_READ_REQ_sum" in e and e.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        e["l1_hit_frac"] = round(1 - e["TCP_TCC_READ_REQ_sum"] / e["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
        e["l2_hit_frac"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
    return e


KERNEL_SOURCES = ("blind_rotate.hpp", "blind_rotate_wide.hpp", "experiment.hpp", "fft512.hpp", "key_switch.hpp", "key_switch_mfma.hpp", "keygen.hpp")


def source_stamp():
    """What the counters were measured on: a digest of the kernel sources and the CMUX-loop instruction counts of the
    build (bench.py refuses to quote an entry whose stamp differs from the tree it is timing)."""
    import glob
    import hashlib
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:  # device code only: host-side changes do not move the counters
        h.update(name.encode())
        h.update(open(os.path.join(root, "rs-tfhe_amd", "csrc", name), "rb").read())
    stamp = {"csrc_sha256": h.hexdigest()[:16]}
    try:
        isa = json.load(open(os.path.join(root, "rs-tfhe_amd", "kernel_isa.json")))
        stamp["valu_per_cmux_step"] = {k: v["valu"] for k, v in isa.items()}
    except (OSError, ValueError, KeyError):
        pass
    return stamp


def dispatch_of_trace_run():
    """The dispatch plan of the traced run (bench.py prints what tfhe_hip_describe_dispatch returned): the host-side
    rules that pick kernels and grids live in tfhe_hip.hip, outside the device-code digest, so the plan itself is
    recorded and bench.py quotes the entry only for the same plan."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(args.pmc)), args.tag + "_trace_bench.json")
    try:
        return json.loads(open(path).readline())["roofline"].get("dispatch")
    except (OSError, ValueError, KeyError, IndexError):
        return None


entry = {
    "tag": args.tag,
    "source": source_stamp(),
    "config": {"params": args.params, "batch": args.batch, "gate": args.gate},
    "dispatch": dispatch_of_trace_run(),
    "kernels_in_trace": sorted({r["kernel"].split("(")[0].replace("void ", "") for r in stats if "tfhe::" in r["kernel"]}),
    "blind_rotate": kernel_entry("k_blind_rotate<"),
    "key_switch": kernel_entry("k_key_switch"),
}
if any("k_ks_digits" in r["kernel"] for r in stats):
    entry["key_switch_digits"] = kernel_entry("k_ks_digits")
json.dump(entry, sys.stdout, indent=1)
print()
