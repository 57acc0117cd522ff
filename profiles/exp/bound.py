"""Should calls larger than #CUs ciphertexts merge too?  A lone caller's time per call of 512 / 1,024 / 4,096 gates through the
direct path (bound 256) and through the front end (bound 4,096), and 4 / 8 concurrent callers of such calls."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rs_tfhe_amd as R
from rs_tfhe_amd import callers
P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=1)
eng = R.Engine(P, 0)
eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2)
rng = np.random.default_rng(3)
M = 32768
ca, cb = sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 1), sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 2)
for bound in (256, 4096):
    eng.set_combining(bound)
    for per in (512, 1024, 4096):
        for T in (1, 4, 8):
            K = max(2, min(6, M // (T * per)))
            n = T * K * per
            if n > M:
                continue
            g = np.zeros(T * K, np.uint8)
            callers.run(eng, callers.OP_GATE, ca[:T * per], cb[:T * per], gates=g[:T], threads=T, calls=1, per_call=per)
            out, secs, ms = callers.run(eng, callers.OP_GATE, ca[:n], cb[:n], gates=g, threads=T, calls=K, per_call=per)
            print(json.dumps({"bound": bound, "per_call": per, "threads": T, "gates_per_s": round(n / secs), "call_ms_median": round(float(np.median(ms)), 2)}), flush=True)
