#!/usr/bin/env python3
"""One call far above BASELINE's batch: 1,048,576 hom_nand in ONE tfhe_hip_batch_gate_dev call (2.9 GB per operand:
byte offsets beyond 2^31 and 2^32 inside the kernels).  4,096 distinct ciphertext pairs tiled 256 times: every tile
must equal the first, the first must equal the CPU path.   python3 profiles/exp/big_batch.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import rs_tfhe_amd as R
from oracle import oracle as O
import test_gpu_parity as T

sk, ck = T.oracle_keys(O, O.SECURITY_128_BIT, seed=1234)
pk = T._cloud_key(ck)
eng = R.Engine(pk.params, 0)
eng.load_cloud_key(pk)
tile, reps = 4096, 256
rng = np.random.default_rng(5)
A, B = rng.integers(0, 2, tile).astype(bool), rng.integers(0, 2, tile).astype(bool)
ca, cb = sk.encrypt_bool(A, 91), sk.encrypt_bool(B, 92)
dev = torch.device("cuda", 0)
ta = torch.from_numpy(ca.view(np.int32)).to(dev).repeat(reps, 1)
tb = torch.from_numpy(cb.view(np.int32)).to(dev).repeat(reps, 1)
to = torch.empty_like(ta)
t0 = time.perf_counter()
eng.batch_gate_dev(R.engine.NAND, ta, tb, to)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
first = to[:tile]
same = all(bool(torch.equal(to[r * tile:(r + 1) * tile], first)) for r in range(1, reps))
want = O.batch_gate(ck, O.GATE_NAND, ca[:512], cb[:512])
ok = np.array_equal(first[:512].cpu().numpy().view(np.uint32), want)
print(f"{tile * reps} bootstraps in one call: {dt:.2f} s = {tile * reps / dt / 1e3:.1f} k/s; all {reps} tiles identical: {same}; first 512 == CPU path: {ok}; "
      f"decrypts: {bool(np.array_equal(sk.decrypt_bool(first.cpu().numpy().view(np.uint32)), ~(A & B)))}")
sys.exit(0 if same and ok else 1)
