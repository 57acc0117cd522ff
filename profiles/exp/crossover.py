#!/usr/bin/env python3
"""Batch-size crossover between the latency kernels (one workgroup of 2l waves per ciphertext, split key switch) and
the batch kernels: per batch size, blind-rotate / key-switch launch times with the latency kernels forced off and on.

    python3 profiles/exp/crossover.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(wide_max, ks_split_max):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    sk = R.SecretKey.new(P, seed=1)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2)
    dev = torch.device("cuda", 0)
    for B in (1, 64, 256, 257, 512, 513, 768, 769, 1024, 1536, 2048):
        bits = np.random.default_rng(B).integers(0, 2, B).astype(bool)
        c = torch.from_numpy(sk.encrypt_bool(bits, seed=3).view(np.int32)).to(dev)
        o = torch.empty_like(c)
        eng.batch_gate_dev(R.engine.NAND, c, c, o)
        torch.cuda.synchronize()
        eng.kernel_times()
        eng.set_profiling(True)
        for _ in range(3):
            eng.batch_gate_dev(R.engine.NAND, c, c, o)
        torch.cuda.synchronize()
        eng.set_profiling(False)
        kt = eng.kernel_times()
        ok = bool(np.array_equal(sk.decrypt_bool(o.cpu().numpy().view(np.uint32)), ~bits))
        print(f"wide_max={wide_max:6d} ks_split_max={ks_split_max:6d} B={B:5d}: blind_rotate {kt['blind_rotate_ms'] / 3:8.2f} ms  "
              f"key_switch {kt['key_switch_ms'] / 3:7.3f} ms  ok={ok}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for wm, ks in ((0, 0), (100000, 100000)):
            env = dict(os.environ, TFHE_HIP_WIDE_MAX=str(wm), TFHE_HIP_KS_SPLIT_MAX=str(ks), TFHE_HIP_PAIR_MAX="0")  # numeric overrides: experiment builds only (TFHE_HIP_LIB=libtfhe_v_*.so)
            subprocess.run([sys.executable, os.path.abspath(__file__), str(wm), str(ks)], env=env)
