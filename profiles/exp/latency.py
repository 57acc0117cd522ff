#!/usr/bin/env python3
"""Single-gate latency: `Gates::nand` on ONE pair of ciphertexts (BASELINE configs[0]'s shape, benches/gate_benchmarks.rs:12-20)
through the host API, device-resident through the *_dev API, and for small batches; with the launch time of the two
kernels (HIP events), the shader clock the latency kernel ran at and its cycles per CMUX step.

    python3 profiles/exp/latency.py [--params SECURITY_128_BIT] [--reps 20]
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
    ap.add_argument("--params", default="SECURITY_128_BIT")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--counts", default="1,2,16,64,256,512")
    args = ap.parse_args()
    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.PARAM_SETS[args.params]
    sk = R.SecretKey.new(P, seed=2024)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    dev = torch.device("cuda", 0)
    rows = []
    for count in [int(c) for c in args.counts.split(",")]:
        rng = np.random.default_rng(count)
        A, B = rng.integers(0, 2, count).astype(bool), rng.integers(0, 2, count).astype(bool)
        ca, cb = sk.encrypt_bool(A, seed=1), sk.encrypt_bool(B, seed=2)
        out = eng.batch_gate(R.engine.NAND, ca, cb)
        assert np.array_equal(sk.decrypt_bool(out), ~(A & B))
        t0 = time.perf_counter()
        for _ in range(args.reps):
            eng.batch_gate(R.engine.NAND, ca, cb)
        host_ms = (time.perf_counter() - t0) / args.reps * 1e3
        ta, tb = torch.from_numpy(ca.view(np.int32)).to(dev), torch.from_numpy(cb.view(np.int32)).to(dev)
        to = torch.empty_like(ta)
        eng.batch_gate_dev(R.engine.NAND, ta, tb, to)
        torch.cuda.synchronize()
        eng.kernel_times()
        eng.clock_sample()
        eng.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            eng.batch_gate_dev(R.engine.NAND, ta, tb, to)
            torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) / args.reps * 1e3
        kt = eng.kernel_times()
        clk = eng.clock_sample()
        eng.set_profiling(False)
        L = max(kt["blind_rotate_launches"], 1)
        rows.append({
            "count": count, "host_api_ms": round(host_ms, 3), "dev_api_ms": round(dev_ms, 3),
            "blind_rotate_ms": round(kt["blind_rotate_ms"] / L, 3), "key_switch_ms": round(kt["key_switch_ms"] / max(kt["key_switch_launches"], 1), 3),
            "shader_mhz": round(clk["shader_mhz"], 1),
            "cycles_per_cmux_step": round(clk["shader_cycles"] / L / count / P.n, 1) if clk["shader_cycles"] else None,
        })
        print(json.dumps(rows[-1]), flush=True)


def stamps(params="SECURITY_128_BIT"):
    """With a -DTFHE_EXPERIMENT -DTFHE_LAT_STAMPS build (TFHE_HIP_LIB): cycles between the phase stamps of step n/2."""
    import ctypes as C

    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.PARAM_SETS[params]
    sk = R.SecretKey.new(P, seed=2024)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    ca, cb = sk.encrypt_bool([True], seed=1), sk.encrypt_bool([False], seed=2)
    eng.batch_gate(R.engine.NAND, ca, cb)
    eng.set_profiling(True)
    eng.batch_gate(R.engine.NAND, ca, cb)
    buf = (C.c_uint64 * 128)()
    fn = eng._lib.tfhe_hip_experiment_diag
    fn.argtypes, fn.restype = [C.c_void_p, C.c_void_p, C.c_size_t], C.c_int
    assert fn(eng._ctx, buf, 128) == 0
    w = np.array(buf[:], dtype=np.uint64)
    names = ["start", "digits", "fwd_fft", "products+keyloads", "barrier1", "partial_sum", "inv_fft", "acc_update", "barrier2"]
    for wave in range(2 * P.l):
        st = w[8 + wave * 16: 8 + wave * 16 + 9].astype(np.int64)
        d = {names[i]: int(st[i] - st[i - 1]) for i in range(1, 9) if st[i] and st[i - 1]}
        print(json.dumps({"wave": wave, "step_total": int(st[8] - st[0]) if st[8] and st[0] else None, **d}), flush=True)


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        stamps()
        sys.exit(0)
    main()
