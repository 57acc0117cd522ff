"""Interactive calls beside bulk work: one thread keeps 65,536-ciphertext batches running on the context's own stream
(330 ms each); another makes one-ciphertext host calls (the front end's lane) and times each.  With the lane's stream at
normal and at the highest priority (TFHE_HIP_LANE_PRIORITY in an experiment build), and on an idle GPU for reference.
    TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$PWD/rs-tfhe_amd/libtfhe_v_comb.so python3 profiles/exp/mixed_load.py"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rs_tfhe_amd as R  # noqa: E402

P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=2024)
rng = np.random.default_rng(3)
B = 65536
ca = sk.encrypt_bool(rng.integers(0, 2, B).astype(bool), 1)
cb = sk.encrypt_bool(rng.integers(0, 2, B).astype(bool), 2)
ta, tb = (torch.from_numpy(x.view(np.int32)).cuda() for x in (ca, cb))
to = torch.empty_like(ta)
for prio in ("0", "1"):
    os.environ["TFHE_HIP_LANE_PRIORITY"] = prio
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    want = eng.batch_gate(0, ca[:64], cb[:64])
    for bulk_count in (0, 65536, 2048):
        stop = threading.Event()
        steps = [0]

        def bulk():
            while not stop.is_set():
                eng.batch_gate_dev(0, ta[:bulk_count], tb[:bulk_count], to[:bulk_count])
                eng.synchronize()
                steps[0] += 1

        th = threading.Thread(target=bulk) if bulk_count else None
        if th:
            th.start()
            time.sleep(0.5)
        lat, ok = [], True
        t_end = time.time() + 4.0
        i = 0
        while time.time() < t_end and len(lat) < 400:
            k = i % 64
            t0 = time.perf_counter()
            out = eng.batch_gate(0, ca[k:k + 1], cb[k:k + 1])
            lat.append((time.perf_counter() - t0) * 1e3)
            ok = ok and bool(np.array_equal(out[0], want[k]))
            i += 1
            time.sleep(0.003)
        stop.set()
        if th:
            th.join()
        lat = np.array(lat)
        print(json.dumps({"lane_priority": "highest" if prio == "1" else "normal", "bulk_batch": bulk_count, "bulk_steps": steps[0], "calls": len(lat),
                          "call_ms_median": round(float(np.median(lat)), 2), "p90": round(float(np.percentile(lat, 90)), 2), "max": round(float(lat.max()), 2),
                          "same_bits": ok}), flush=True)
    eng.close()
