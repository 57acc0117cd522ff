#!/usr/bin/env python3
"""k_blind_rotate_pair (two ciphertexts per eight-wave workgroup) against the other blind-rotation kernels: same bits,
and launch times by batch size.   python3 profiles/exp/pair_check.py [--time]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(mode):
    sys.path.insert(0, ROOT)
    import hashlib

    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.PARAM_SETS[os.environ.get("PAIR_PARAMS", "SECURITY_128_BIT")]
    sk = R.SecretKey.new(P, seed=1)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2)
    if mode == "bits":
        for B in (1, 2, 5, 40, 301):
            bits = np.random.default_rng(B).integers(0, 2, B).astype(bool)
            ca, cb = sk.encrypt_bool(bits, seed=3), sk.encrypt_bool(~bits, seed=4)
            outs = [eng.batch_gate(R.engine.NAND, ca, cb), eng.batch_blind_rotate(ca), eng.batch_bootstrap(ca, keyswitch=False),
                    eng.batch_gates_mixed(np.arange(B, dtype=np.uint8) % 10, ca, cb)]
            h = hashlib.sha256(b"".join(np.ascontiguousarray(o).tobytes() for o in outs)).hexdigest()[:16]
            print(f"B={B:4d} digest {h} nand_ok={bool(np.array_equal(sk.decrypt_bool(outs[0]), np.ones(B, bool)))}", flush=True)
        return
    dev = torch.device("cuda", 0)
    for B in (256, 257, 512, 768, 1024, 1100, 1280, 1500, 1536, 2048, 2200, 2500, 3072, 3300, 4096, 4200):
        bits = np.random.default_rng(B).integers(0, 2, B).astype(bool)
        c = torch.from_numpy(sk.encrypt_bool(bits, seed=3).view(np.int32)).to(dev)
        o = torch.empty_like(c)
        eng.batch_gate_dev(R.engine.NAND, c, c, o)
        torch.cuda.synchronize()
        eng.kernel_times(); eng.clock_sample()
        eng.set_profiling(True)
        for _ in range(3):
            eng.batch_gate_dev(R.engine.NAND, c, c, o)
        torch.cuda.synchronize()
        eng.set_profiling(False)
        kt = eng.kernel_times(); clk = eng.clock_sample()
        ok = bool(np.array_equal(sk.decrypt_bool(o.cpu().numpy().view(np.uint32)), ~bits))
        print(f"B={B:5d}: blind_rotate {kt['blind_rotate_ms'] / 3:8.2f} ms  {clk['shader_mhz']:.0f} MHz  ok={ok}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        mode = "time" if "--time" in sys.argv else "bits"
        for name, env in (("pair kernel everywhere", {"TFHE_HIP_PAIR_LO": "0", "TFHE_HIP_PAIR_MAX": "1000000", "TFHE_HIP_WIDE_MAX": "0"}),
                          ("shipped dispatch", {}),
                          ("no pair kernel", {"TFHE_HIP_PAIR_MAX": "0"})):
            print("#", name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=dict(os.environ, **env))
