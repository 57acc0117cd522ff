#!/usr/bin/env python3
"""Key switch alone: identity_key_switching (trgsw.rs:332-360) of a batch of random level-1 ciphertexts through
tfhe_hip_batch_identity_key_switch, timed with the library's HIP events, with the board power sampled beside it
and a digest of the outputs (every correct kernel variant prints the same one).  Variants are chosen by the
TFHE_HIP_KS_* environment variables or by TFHE_HIP_LIB (a differently built library).

    python3 profiles/exp/ks_only.py [--batch 65536] [--params SECURITY_128_BIT] [--reps 5]
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--params", default="SECURITY_128_BIT")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np

    import rs_tfhe_amd as R

    P = R.params.PARAM_SETS[args.params]
    sk = R.SecretKey.new(P, seed=2024)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    rng = np.random.default_rng(77)
    lv1 = rng.integers(0, 2**32, (args.batch, 1025), dtype=np.uint64).astype(np.uint32)
    out = eng.batch_identity_key_switch(lv1)
    eng.kernel_times()
    eng.set_profiling(True)
    pfiles = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
    watts, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            best = 0.0
            for f in pfiles:
                try:
                    best = max(best, float(open(f).read()) * 1e-6)
                except (OSError, ValueError):
                    pass
            watts.append(best)
            time.sleep(0.005)

    th = threading.Thread(target=sample)
    th.start()
    for _ in range(args.reps):
        out = eng.batch_identity_key_switch(lv1)
    stop.set()
    th.join()
    kt = eng.kernel_times()
    clk = eng.key_switch_clock_sample()
    print(json.dumps({
        "params": args.params, "batch": args.batch,
        "env": {k: v for k, v in os.environ.items() if k.startswith("TFHE_HIP_")},
        "key_switch_ms": round(kt["key_switch_ms"] / max(kt["key_switch_launches"], 1), 4),
        "launches": kt["key_switch_launches"],
        "shader_mhz": round(clk["shader_mhz"], 1), "shader_cycles_per_workgroup_launch": clk["shader_cycles"] / max(kt["key_switch_launches"], 1),
        "max_board_w": round(max(watts), 1) if watts else None,
        "digest": hashlib.sha256(out.tobytes()).hexdigest()[:16],
    }))


if __name__ == "__main__":
    main()
