import sys; sys.path.insert(0, '.')
import numpy as np, torch
import rs_tfhe_amd as R
P = R.params.SECURITY_80_BIT
sk = R.SecretKey.new(P, seed=1)
pool = R.Pool(P, [0, 0])
pool.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2)
ca = sk.encrypt_bool(np.ones(1, bool), seed=3)
ta = torch.from_numpy(ca.view(np.int32)).to("cuda:0")
to = torch.empty_like(ta)
print("calling", flush=True)
pool.batch_gate_dev(0, ta, ta, to, home=0)
pool.synchronize(); torch.cuda.synchronize()
print("ok", sk.decrypt_bool(to.cpu().numpy().view(np.uint32)))
