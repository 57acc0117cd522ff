"""Soak of the combining front end: a C++ team of 96 threads making one-ciphertext gate calls in a loop, beside Python
threads that make every other kind of small call on two key views, large batches on the context's own stream, and key
reloads on a third view -- for --seconds.  Every result is compared with the same operation done as ONE plain batch call
after the storm (the batch path is what the parity suite holds to the CPU checker).
    python profiles/exp/soak_front_end.py --seconds 120"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rs_tfhe_amd as R  # noqa: E402
from rs_tfhe_amd import callers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
args = ap.parse_args()
P = R.params.SECURITY_128_BIT
N = 1024
sk1, sk2 = R.SecretKey.new(P, seed=1), R.SecretKey.new(P, seed=2)
eng = R.Engine(P, 0)
eng.gen_cloud_key(sk1.key_lv0, sk1.key_lv1, seed=11)
v2 = eng.new_key_view()
v2.gen_cloud_key(sk2.key_lv0, sk2.key_lv1, seed=12)
v3 = eng.new_key_view()
rng = np.random.default_rng(9)
M = 96 * 40
ca, cb = sk1.encrypt_bool(rng.integers(0, 2, M).astype(bool), 1), sk1.encrypt_bool(rng.integers(0, 2, M).astype(bool), 2)
gates = rng.integers(0, 10, M).astype(np.uint8)
da, db, dc = (sk2.encrypt_bool(rng.integers(0, 2, 64).astype(bool), 3 + i) for i in range(3))
tv = rng.integers(0, 2**32, (64, 2, N), dtype=np.uint64).astype(np.uint32)
big_a, big_b = sk1.encrypt_bool(rng.integers(0, 2, 3000).astype(bool), 7), sk1.encrypt_bool(rng.integers(0, 2, 3000).astype(bool), 8)
import torch  # noqa: E402

bulk_a, bulk_b = (torch.from_numpy(np.tile(x, (7, 1)).view(np.int32)).cuda() for x in (big_a, big_b))  # 21,000: cut into chunks while small calls arrive
bulk_out = torch.empty_like(bulk_a)
bound = eng.combine_stats()["max_count"]
eng.set_combining(0)  # the references: plain batch calls, front end off
ref_gates = eng.batch_gates_mixed(gates, ca, cb)
ref = {"mixed": v2.batch_gates_mixed(gates[:64], da, db), "boot": v2.batch_bootstrap(da), "boot_nks": v2.batch_bootstrap(da, keyswitch=False),
       "lut": v2.batch_bootstrap(da, tv), "mux": v2.batch_mux(da, db, dc, naive=False), "mux_naive": v2.batch_mux(da, db, dc, naive=True),
       "big": eng.batch_gate(0, big_a, big_b)}
ref["bulk"] = np.tile(ref["big"], (7, 1))
eng.set_combining(bound)
stop = time.time() + args.seconds
bad, counts, lock = [], {}, threading.Lock()


def note(kind, ok):
    with lock:
        counts[kind] = counts.get(kind, 0) + 1
        if not ok:
            bad.append(kind)


def team():
    while time.time() < stop:
        out, _, _ = callers.run(eng, callers.OP_GATE, ca, cb, gates=gates, threads=96, calls=40)
        note("team_rounds", bool(np.array_equal(out, ref_gates)))


def small(seed):
    r = np.random.default_rng(seed)
    while time.time() < stop:
        k = int(r.integers(0, 6))
        lo = int(r.integers(0, 56))
        n = int(r.integers(1, 9))
        s = slice(lo, lo + n)
        if k == 0:
            ok = np.array_equal(v2.batch_gates_mixed(gates[:64][s], da[s], db[s]), ref["mixed"][s])
        elif k == 1:
            ok = np.array_equal(v2.batch_bootstrap(da[s]), ref["boot"][s])
        elif k == 2:
            ok = np.array_equal(v2.batch_bootstrap(da[s], keyswitch=False), ref["boot_nks"][s])
        elif k == 3:
            ok = np.array_equal(v2.batch_bootstrap(da[s], tv[s]), ref["lut"][s])
        elif k == 4:
            ok = np.array_equal(v2.batch_mux(da[s], db[s], dc[s], naive=False), ref["mux"][s])
        else:
            ok = np.array_equal(v2.batch_mux(da[s], db[s], dc[s], naive=True), ref["mux_naive"][s])
        note(("mixed", "boot", "boot_nks", "lut", "mux", "mux_naive")[k], bool(ok))


def big():
    while time.time() < stop:
        note("big", bool(np.array_equal(eng.batch_gate(0, big_a, big_b), ref["big"])))
        time.sleep(0.05)


def bulk():
    while time.time() < stop:
        eng.batch_gate_dev(0, bulk_a, bulk_b, bulk_out)
        eng.synchronize()
        note("bulk_21000_dev", bool(np.array_equal(bulk_out.cpu().numpy().view(np.uint32), ref["bulk"])))
        time.sleep(0.02)


def reload():
    i = 0
    while time.time() < stop:
        v3.gen_cloud_key(sk2.key_lv0, sk2.key_lv1, seed=12)  # the same key as v2: results must equal v2's
        note("reload", bool(np.array_equal(v3.batch_bootstrap(da[:3]), ref["boot"][:3])))
        i += 1
        time.sleep(0.2)


threads = [threading.Thread(target=team)] + [threading.Thread(target=small, args=(100 + i,)) for i in range(6)] + [threading.Thread(target=big), threading.Thread(target=bulk), threading.Thread(target=reload)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print(json.dumps({"seconds": args.seconds, "checked": counts, "team_gates": counts.get("team_rounds", 0) * M, "wrong": bad, "stats": eng.combine_stats()}))
