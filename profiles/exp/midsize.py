"""Round 6, one bounded experiment (verdict item 5): for 768 < count <= 2,048 the dispatch plan issues up to two
launches back to back on one stream.  Do the two parts OVERLAP when they go to two streams?  And does a forced cut
(batch[0, 512) beside pair[512, 1024), two half grids of the batch kernel, ...) beat the plan?
Needs an experiment build (TFHE_HIP_BR_OVERLAP / TFHE_HIP_BR_SPLIT are read only there):
    bash profiles/exp/build_variants.sh comb ""
    TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$PWD/rs-tfhe_amd/libtfhe_v_comb.so python3 profiles/exp/midsize.py"""
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rs_tfhe_amd as R  # noqa: E402

P = R.params.SECURITY_128_BIT
sk = R.SecretKey.new(P, seed=2024)
rng = np.random.default_rng(3)
M = 2048
ca = sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 1)
cb = sk.encrypt_bool(rng.integers(0, 2, M).astype(bool), 2)
ta, tb = (torch.from_numpy(x.view(np.int32)).cuda() for x in (ca, cb))
to = torch.empty_like(ta)
VARIANTS = [("plan, one stream", {}), ("plan, two streams", {"TFHE_HIP_BR_OVERLAP": "1"})]
for at, k0, k1, name in ((512, 0, 2, "batch[0,512) | pair"), (512, 0, 0, "batch[0,512) | batch"), (256, 1, 0, "single[0,256) | batch"),
                         (512, 2, 0, "pair[0,512) | batch"), (1024, 0, 2, "batch[0,1024) | pair"), (1024, 0, 0, "batch[0,1024) | batch")):
    for ov in ("0", "1"):
        VARIANTS.append((f"{name}, {'two streams' if ov == '1' else 'one stream'}", {"TFHE_HIP_BR_SPLIT": f"{at}:{k0}:{k1}", "TFHE_HIP_BR_OVERLAP": ov}))
ref = {}
for name, env in VARIANTS:
    for k in ("TFHE_HIP_BR_OVERLAP", "TFHE_HIP_BR_SPLIT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    row = {"variant": name}
    for count in (768, 1024, 1100, 1280, 1536, 2048):
        at = int(env.get("TFHE_HIP_BR_SPLIT", "0:0:0").split(":")[0])
        if at and count <= at:
            continue
        for _ in range(3):
            eng.batch_gate_dev(0, ta[:count], tb[:count], to[:count])
        torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            eng.batch_gate_dev(0, ta[:count], tb[:count], to[:count])
            eng.synchronize()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        dig = hashlib.sha256(to[:count].cpu().numpy().tobytes()).hexdigest()[:12]
        ref.setdefault(count, dig)
        row[str(count)] = round(statistics.median(ts), 2)
        if dig != ref[count]:
            row[f"{count}_DIGEST_DIFFERS"] = dig
    eng.close()
    print(json.dumps(row), flush=True)
