#!/usr/bin/env python3
"""Fuzz of the run-time parametric engine: RANDOM parameter shapes (n, l, bgbit, basebit, t) inside what
tfhe_hip_ctx_create accepts and inside the exact-product regime (so that every word has one right answer), random batch
sizes, every blind-rotation kernel, random entry points -- gates, mixed gate codes, bootstrap with a shared / per-ciphertext
test vector, with and without the key switch, linear combinations fused into the bootstrap, mux in both forms, the
device-resident pool calls over 1..8 members with a random home -- every output word against the CPU oracle.

    python3 profiles/exp/fuzz_shapes.py [--seed 1] [--trials 40] [--max-count 48]

Not part of the test suite (its shapes are random): a bug it finds becomes a fixed test.  Exit code 1 if anything differs."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
N = 1024


def random_shape(rng):
    """(n, l, bgbit, basebit, t): exact regime bgbit + log2(2l) < 12 (2l * N * Bg/2 * 2^31 < 2^52), l * bgbit <= 32,
    basebit * t <= 31, key-switching key <= ~256 MB."""
    import numpy as np

    while True:
        l = int(rng.integers(1, 4))
        bgmax = {1: 10, 2: 9, 3: 9}[l]
        bgbit = int(rng.integers(2, bgmax + 1))
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 511, 512, 513, 550, 700, 701, 820, 1023, 1024, 1025, 1071, 1160,
                            1279, int(rng.integers(1, 1280)), int(rng.integers(1, 1280))]))
        basebit = int(rng.integers(1, 8))
        t = int(rng.integers(1, max(2, min(12, 31 // basebit) + 1)))
        if basebit * t > 31:
            continue
        ksk_bytes = N * t * (1 << basebit) * (n + 1) * 4
        if ksk_bytes > 256e6:
            continue
        return n, l, bgbit, basebit, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--max-count", type=int, default=48)
    args = ap.parse_args()
    import numpy as np
    import torch

    import rs_tfhe_amd as R
    from oracle import oracle as O
    from rs_tfhe_amd.params import SecurityParams

    O.build()
    rng = np.random.default_rng(args.seed)
    total = bad = 0
    t0 = time.time()
    failures = []
    for trial in range(args.trials):
        n, l, bgbit, basebit, t = random_shape(rng)
        op = O.Params(f"FUZZ_{trial}", n, l, bgbit, basebit, t, 2.0e-5, 2.0e-8)
        sk, ck = O.keygen(op, 5000 + 17 * trial + args.seed)
        pp = SecurityParams(op.name, 0, n, l, bgbit, basebit, t, op.alpha_lv0, op.alpha_lv1)
        pk = R.CloudKey(pp, ck.bootstrapping_key, ck.key_switching_key, ck.decomposition_offset, ck.blind_rotate_testvec)
        br = ["auto", "batch", "single", "pair"][int(rng.integers(0, 4))]
        ks = ["auto", "auto", "mfma", "sliced", "b4", "generic", "split"][int(rng.integers(0, 7))]
        os.environ["TFHE_HIP_BR_KERNEL"] = br
        os.environ["TFHE_HIP_KS_KERNEL"] = ks
        count = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, args.max_count, 65, 130, 400, 700]))
        if n <= 300 and rng.random() < 0.2:  # small n: the oracle is cheap enough for counts beyond the batch kernel's first rounds
            count = int(rng.choice([1025, 1100, 1500, 2049, 2200, 3300]))
        few = count > 64   # large counts: the gate, mixed-gate and key-switch entry points only (the oracle's time)
        nks_ok = n <= N    # sample_extract_index_2 (trlwe.rs:122-136) reads a[n - i]: n <= N, or the reference itself is out of bounds
        a = rng.integers(0, 2**32, (count, n + 1), dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2**32, (count, n + 1), dtype=np.uint64).astype(np.uint32)
        c = rng.integers(0, 2**32, (count, n + 1), dtype=np.uint64).astype(np.uint32)
        tv1 = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
        tvn = rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32)
        codes = rng.integers(0, 11, count).astype(np.uint8)
        gate = int(rng.integers(0, 10))
        checks = []
        try:
            eng = R.Engine(pp, 0)
        except R._capi.TfheHipError as e:
            if "TFHE_HIP_KS_KERNEL" not in str(e) and "TFHE_HIP_BR_KERNEL" not in str(e):
                raise
            # the kernel named cannot run this shape (documented: creation fails): let the library choose
            os.environ["TFHE_HIP_KS_KERNEL"] = ks = "auto"
            if "BR_KERNEL" in str(e):
                os.environ["TFHE_HIP_BR_KERNEL"] = br = "auto"
            eng = R.Engine(pp, 0)
        try:
            eng.load_cloud_key(pk)
            plan = eng.describe_dispatch(count)

            def gates_mixed_want(ks):
                w = np.empty((count, n + 1), np.uint32)
                for g in range(11):
                    m = codes == g
                    if not m.any():
                        continue
                    if g == 10:  # COPY: bootstrap of a itself
                        w[m] = O.batch_bootstrap(ck, a[m], keyswitch=ks)
                    else:
                        prep = np.stack([O.gate_prep(g, x, y, n) for x, y in zip(a[m], b[m])])
                        w[m] = O.batch_bootstrap(ck, prep, keyswitch=ks)
                return w

            checks.append(("gate", eng.batch_gate(gate, a, b), O.batch_gate(ck, gate, a, b)))
            checks.append(("gates_mixed", eng.batch_gates_mixed(codes, a, b), gates_mixed_want(True)))
            if nks_ok and not few:
                checks.append(("gates_mixed_nks", eng.batch_gates_mixed(codes, a, b, keyswitch=False), gates_mixed_want(False)))
                checks.append(("bootstrap per-ct nks", eng.batch_bootstrap(a, testvec=tvn, keyswitch=False),
                               O.batch_bootstrap(ck, a, testvec=tvn, keyswitch=False)))
                checks.append(("mux", eng.batch_mux(a, b, c, naive=False), O.batch_mux(ck, a, b, c, naive=False)))
            if not few:
                checks.append(("bootstrap tv1", eng.batch_bootstrap(a, testvec=tv1), O.batch_bootstrap(ck, a, testvec=tv1)))
                checks.append(("bootstrap per-ct", eng.batch_bootstrap(a, testvec=tvn), O.batch_bootstrap(ck, a, testvec=tvn)))
                checks.append(("blind_rotate", eng.batch_blind_rotate(a), O.batch_blind_rotate(ck, a)))
                checks.append(("mux_naive", eng.batch_mux(a, b, c, naive=True), O.batch_mux(ck, a, b, c, naive=True)))
                lin = (3 * a - 2 * b).astype(np.uint32)
                lin[:, n] += np.uint32(0x01234567)
                checks.append(("lincomb_bootstrap", eng.batch_lincomb_bootstrap(3, a, -2, b, 0x01234567, testvec=tv1),
                               O.batch_bootstrap(ck, lin, testvec=tv1)))
            lv1 = rng.integers(0, 2**32, (count, N + 1), dtype=np.uint64).astype(np.uint32)
            checks.append(("key_switch", eng.batch_identity_key_switch(lv1), O.batch_identity_key_switching(ck, lv1)))
            # device-resident pool call: 1..8 members on device 0, random home
            members = int(rng.integers(1, 9))
            home = int(rng.integers(0, members))
            pool = R.Pool(pp, [0] * members)
            pool.load_cloud_key(pk)
            big = int(rng.choice([count, 257 * members + 3]))
            a2 = rng.integers(0, 2**32, (big, n + 1), dtype=np.uint64).astype(np.uint32)
            b2 = rng.integers(0, 2**32, (big, n + 1), dtype=np.uint64).astype(np.uint32)
            ta, tb = (torch.from_numpy(x.view(np.int32)).to("cuda:0") for x in (a2, b2))
            want_t, got_t = torch.empty_like(ta), torch.full_like(ta, 0x5A5A5A5A)
            eng.batch_gate_dev(gate, ta, tb, want_t)
            pool.batch_gate_dev(gate, ta, tb, got_t, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            idx = np.unique(np.r_[0:min(big, 6), np.linspace(0, big - 1, 12).astype(np.int64)])
            checks.append((f"pool[{members}] home {home} x{big}", got_t.cpu().numpy().view(np.uint32), want_t.cpu().numpy().view(np.uint32)))
            checks.append(("engine _dev vs oracle", want_t.cpu().numpy().view(np.uint32)[idx], O.batch_gate(ck, gate, a2[idx], b2[idx])))
            pool.close()
        except R._capi.TfheHipError as e:
            failures.append((trial, (n, l, bgbit, basebit, t), br, ks, "ERROR", str(e)))
            print(f"trial {trial}: shape {(n, l, bgbit, basebit, t)} br={br} count {count}: ERROR {e}", flush=True)
            eng.close()
            bad += 1
            continue
        eng.close()
        nbad = 0
        for name, got, want in checks:
            total += len(want)
            d = int((np.asarray(got).reshape(len(want), -1) != np.asarray(want).reshape(len(want), -1)).any(axis=1).sum())
            if d:
                nbad += d
                failures.append((trial, (n, l, bgbit, basebit, t), br, ks, name, f"{d} of {len(want)} differ"))
        bad += nbad
        print(f"trial {trial:3d}: shape {(n, l, bgbit, basebit, t)} br={br} ks={ks} count {count} [{plan}]: "
              f"{'ok' if nbad == 0 else str(nbad) + ' DIFFER'}  ({time.time() - t0:.0f} s)", flush=True)
    for f in failures:
        print("FAIL", f, flush=True)
    print(f"TOTAL {total} results compared over {args.trials} shapes, {bad} differ, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
