"""A one-gate call through the Python mirror (Gates().nand: keyed_engine asks tfhe_hip_key_is_loaded first) while another
thread runs 65,536-ciphertext HOST-pointer batches, which hold the context's mutex for their whole duration."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rs_tfhe_amd as R
P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=1)
ck = R.CloudKey.new(sk, seed=2)
rng = np.random.default_rng(3)
B = 65536
ca, cb = sk.encrypt_bool(rng.integers(0, 2, B).astype(bool), 1), sk.encrypt_bool(rng.integers(0, 2, B).astype(bool), 2)
g = R.Gates()
g.nand(ca[0], cb[0], ck)
stop = threading.Event()
def bulk():
    while not stop.is_set():
        R.gates.batch_nand(ca, cb, ck)
th = threading.Thread(target=bulk); th.start(); time.sleep(1.0)
lat = []
for i in range(40):
    t0 = time.perf_counter(); g.nand(ca[i], cb[i], ck); lat.append((time.perf_counter() - t0) * 1e3); time.sleep(0.01)
stop.set(); th.join()
print(json.dumps({"calls": len(lat), "median_ms": round(float(np.median(lat)), 1), "p90_ms": round(float(np.percentile(lat, 90)), 1), "max_ms": round(max(lat), 1)}))
