#!/usr/bin/env python3
"""Dependent-gate circuits on the latency path: the reference's examples/add_two_numbers.rs shape (ripple-carry adder,
gate by gate) as a levelised device-resident circuit.  Wall time per addition and per level, for B additions at once.

    python3 profiles/exp/adder_latency.py [--bits 16] [--batches 1,16,64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, default=16)
    ap.add_argument("--batches", default="1,16,64")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    sk = R.SecretKey.new(P, seed=2024)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    nb = args.bits
    c = R.circuit.Circuit(2 * nb + 1)
    sums, carry = c.add(list(range(nb)), list(range(nb, 2 * nb)), 2 * nb)
    depth = len(c.levels())
    dev = torch.device("cuda", 0)
    for B in [int(x) for x in args.batches.split(",")]:
        rng = np.random.default_rng(B)
        x, y = rng.integers(0, 1 << nb, B), rng.integers(0, 1 << nb, B)
        bits = np.concatenate([[(x >> i) & 1 for i in range(nb)], [(y >> i) & 1 for i in range(nb)], [np.zeros(B, int)]]).astype(bool)
        inp = np.stack([sk.encrypt_bool(bits[i], seed=100 + i) for i in range(2 * nb + 1)])
        t = torch.from_numpy(inp.view(np.int32)).to(dev)
        wires = c.run_dev(eng, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            wires = c.run_dev(eng, t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        w = wires.cpu().numpy().view(np.uint32)
        got = sum(sk.decrypt_bool(w[s]).astype(np.int64) << i for i, s in enumerate(sums)) + (sk.decrypt_bool(w[carry]).astype(np.int64) << nb)
        print(json.dumps({"bits": nb, "additions": B, "gates": len(c.gates), "levels": depth, "ms_per_run": round(ms, 2),
                          "ms_per_level": round(ms / depth, 3), "correct": bool(np.array_equal(got, x + y))}), flush=True)


if __name__ == "__main__":
    main()
