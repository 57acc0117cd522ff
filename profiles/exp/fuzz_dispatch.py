#!/usr/bin/env python3
"""One-off fuzz of the host's kernel dispatch: batch sizes on and around every crossover (latency kernels 256 / 512 /
768, split -> matrix-core key switch 512, key-switch row padding 128) x the bit-exact parameter sets x random gates,
every output word against the CPU oracle.   python3 profiles/exp/fuzz_dispatch.py [--seed 1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np

    import rs_tfhe_amd as R
    from oracle import oracle as O
    import test_gpu_parity as T

    counts = [1, 2, 3, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 640, 767, 768, 769, 1023, 1024, 1025, 1100, 1500, 2047, 2049, 2200]
    rng = np.random.default_rng(args.seed)
    total = bad = 0
    t0 = time.time()
    for setname in ("SECURITY_128_BIT", "SECURITY_80_BIT", "SECURITY_110_BIT", "SECURITY_UINT1"):
        sk, ck = T.oracle_keys(O, getattr(O, setname), seed=900 + args.seed)
        pk = T._cloud_key(ck)
        eng = R.Engine(pk.params, 0)
        eng.load_cloud_key(pk)
        for count in counts:
            op = int(rng.integers(0, 10))
            A, B = rng.integers(0, 2, count).astype(bool), rng.integers(0, 2, count).astype(bool)
            ca, cb = sk.encrypt_bool(A, int(rng.integers(1 << 30))), sk.encrypt_bool(B, int(rng.integers(1 << 30)))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                got, want = eng.batch_gate(op, ca, cb), O.batch_gate(ck, op, ca, cb)
            elif mode == 1:
                codes = rng.integers(0, 10, count).astype(np.uint8)
                got = eng.batch_gates_mixed(codes, ca, cb)
                want = np.empty_like(got)
                for g in range(10):
                    m = codes == g
                    if m.any():
                        want[m] = O.batch_gate(ck, g, ca[m], cb[m])
            else:
                got, want = eng.batch_bootstrap(ca, keyswitch=True), O.batch_bootstrap(ck, ca, keyswitch=True)
            nbad = int((got != want).any(axis=1).sum())
            bad += nbad
            total += count
            print(f"{setname:18s} count {count:5d} mode {mode} op {op}: {'ok' if nbad == 0 else f'{nbad} DIFFER'}", flush=True)
        eng.close()
    print(f"TOTAL {total} bootstraps, {bad} differ, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
