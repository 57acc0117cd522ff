"""Where a leader's time goes (tfhe_hip_get_combine_stats: linger / pack / gpu / unpack), per team size."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rs_tfhe_amd as R
from rs_tfhe_amd import callers
P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=2024)
eng = R.Engine(P, 0)
eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
rng = np.random.default_rng(5)
M = 16384
ca, cb = sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 1), sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 2)
for T in (1, 8, 64, 128, 256, 512, 1024):
    K = max(3, min(M // T, int(0.5 / (2.4e-3 * max(1.0, T / 256.0)))))
    n = T * K
    g = np.zeros(n, np.uint8)
    callers.run(eng, callers.OP_GATE, ca[:T * 2], cb[:T * 2], gates=g[:T * 2], threads=T, calls=2)
    eng.combine_stats()
    out, secs, ms = callers.run(eng, callers.OP_GATE, ca[:n], cb[:n], gates=g, threads=T, calls=K)
    st = eng.combine_stats()
    L = max(1, st["launches"])
    print(json.dumps({"threads": T, "gates_per_s": round(n / secs), "launches": st["launches"], "reqs_per_launch": round(st["requests"] / L, 1),
                      "round_ms": round(secs * 1e3 / L, 3), "linger_ms": round(st["linger_us"] / L / 1e3, 3), "pack_ms": round(st["pack_us"] / L / 1e3, 3),
                      "gpu_ms": round(st["gpu_us"] / L / 1e3, 3), "unpack_ms": round(st["unpack_us"] / L / 1e3, 3),
                      "other_ms": round((secs * 1e6 - st["linger_us"] - st["pack_us"] - st["gpu_us"] - st["unpack_us"]) / L / 1e3, 3)}), flush=True)
