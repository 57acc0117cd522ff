import sys, json, time
sys.path.insert(0, ".")
import numpy as np
import rs_tfhe_amd as R
P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=2024)
eng = R.Engine(P, 0)
eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
for count in (1, 256):
    ca, cb = sk.encrypt_bool(np.ones(count, bool), seed=1), sk.encrypt_bool(np.zeros(count, bool), seed=2)
    eng.batch_gate(R.engine.NAND, ca, cb)
    eng.kernel_times(); eng.clock_sample(); eng.set_profiling(True)
    for _ in range(10): eng.batch_gate(R.engine.NAND, ca, cb)
    kt = eng.kernel_times(); clk = eng.clock_sample(); eng.set_profiling(False)
    L = kt["blind_rotate_launches"]
    print(sys.argv[1], count, round(kt["blind_rotate_ms"]/L,3), "ms", round(clk["shader_cycles"]/L/count/P.n), "cycles/step", flush=True)
