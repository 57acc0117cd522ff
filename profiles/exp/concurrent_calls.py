"""Aggregate rate of T host threads making one-ciphertext calls on ONE context (combine.hpp), front end on and off.
    python profiles/exp/concurrent_calls.py [--threads 1,8,64,256] [--pool 0,0]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rs_tfhe_amd as R  # noqa: E402
from rs_tfhe_amd import callers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", default="1,2,8,16,64,128,256,512,1024")
ap.add_argument("--pool", default=None)
ap.add_argument("--params", default="SECURITY_128_BIT")
ap.add_argument("--seconds", type=float, default=0.6)
ap.add_argument("--off", action="store_true", help="also the same calls with the front end off (8 threads)")
args = ap.parse_args()
P = R.params.PARAM_SETS[args.params]
sk = R.SecretKey.new(P, seed=2024)
if args.pool:
    target = R.Pool(P, [int(d) for d in args.pool.split(",")])
else:
    target = R.Engine(P, 0)
target.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
rng = np.random.default_rng(5)
M = 16384
A, B = rng.integers(0, 2, M).astype(bool), rng.integers(0, 2, M).astype(bool)
ca, cb = sk.encrypt_bool(A, 1), sk.encrypt_bool(B, 2)
for T in [int(t) for t in args.threads.split(",")]:
    est = 2.4e-3 * max(1.0, T / 256.0)  # seconds per round of T calls
    K = max(3, min(M // T, int(args.seconds / est)))
    n = T * K
    gates = np.zeros(n, np.uint8)
    target.combine_stats()
    w = T * min(K, 3)
    callers.run(target, callers.OP_GATE, ca[:w], cb[:w], gates=gates[:w], threads=T, calls=min(K, 3))  # warm (lanes, arenas)
    target.combine_stats()
    out, secs, ms = callers.run(target, callers.OP_GATE, ca[:n], cb[:n], gates=gates, threads=T, calls=K)
    st = target.combine_stats()
    ok = bool(np.array_equal(sk.decrypt_bool(out), ~(A[:n] & B[:n])))
    print(json.dumps({"threads": T, "calls_per_thread": K, "gates_per_s": round(n / secs, 1), "call_ms_median": round(float(np.median(ms)), 3),
                      "call_ms_p99": round(float(np.percentile(ms, 99)), 3), "decrypt_ok": ok, "stats": st}), flush=True)
if args.off:
    target.set_combining(0)
    T, K = 8, 25
    out, secs, ms = callers.run(target, callers.OP_GATE, ca[:T * K], cb[:T * K], gates=np.zeros(T * K, np.uint8), threads=T, calls=K)
    print(json.dumps({"front_end": "off", "threads": T, "gates_per_s": round(T * K / secs, 1), "call_ms_median": round(float(np.median(ms)), 3)}))
