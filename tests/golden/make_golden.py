"""Generates the committed fixtures in tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference (a Rust crate) cannot be run in this environment and holds no
golden vectors of its own, so these fixtures are the
oracle's outputs on deterministic inputs: they freeze the restated semantics
(so the oracle cannot drift silently) and give the GPU tests key material and
expected outputs that do not depend on any generator at test time.
Fixtures are data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

N = 1024


def kat_inputs():
    """Deterministic inputs lifted from the reference's own tests."""
    i = np.arange(N, dtype=np.uint64)
    kats = {}
    # src/fft/processors.rs:850-855
    kats["consistency_a"] = ((i * 12345) % (1 << 20)).astype(np.uint32)
    kats["consistency_b"] = ((i * 54321) % (1 << 20)).astype(np.uint32)
    # src/fft/processors.rs:809-813
    kats["dense_a"] = ((i * 1234567) % (1 << 20)).astype(np.uint32)
    kats["dense_b"] = ((i * 7654321) % (1 << 20)).astype(np.uint32)
    # src/fft/processors.rs:783-786
    sa = np.zeros(N, np.uint32)
    sb = np.zeros(N, np.uint32)
    sa[::20] = 1 << 28
    sb[::20] = 1 << 27
    kats["sparse_a"], kats["sparse_b"] = sa, sb
    # src/fft/klemsa.rs:187-189
    kl = np.zeros(N, np.uint32)
    kl[0] = 1 << 31
    kl[5] = 1 << 30
    kats["klemsa_roundtrip"] = kl
    # src/fft/mod.rs:166-167
    de = np.zeros(N, np.uint32)
    de[0] = 1000
    kats["delta"] = de
    return kats


def stage_vectors():
    rng = np.random.default_rng(20240601)
    out = {}
    k = kat_inputs()
    for name, v in k.items():
        out["kat_" + name] = v
    for pair in ("consistency", "dense", "sparse"):
        out[f"kat_{pair}_expected"] = O.negacyclic_schoolbook(k[pair + "_a"], k[pair + "_b"])
    # constants (SURVEY 8c iii)
    out["const_f64_to_torus_in"] = np.array([0.125, -0.125, 0.25, -0.25, 0.5, -0.5, 1.0 / 64, 1.75, -1.75, 0.0])
    out["const_f64_to_torus_out"] = np.array([O.f64_to_torus(float(x)) for x in out["const_f64_to_torus_in"]], np.uint32)
    out["const_decomp_offset_lbg"] = np.array([[3, 6], [2, 10], [1, 18], [1, 22], [1, 23]], np.int32)
    out["const_decomp_offset"] = np.array(
        [O.gen_decomposition_offset(int(l), int(bg)) for l, bg in out["const_decomp_offset_lbg"]], np.uint32
    )
    # rotation (trgsw.rs:307-330) incl. k = 0, N, 2N
    poly = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    ks = np.array([0, 1, 5, 511, 512, 1023, 1024, 1025, 1500, 2047, 2048], np.int32)
    out["rot_in"] = poly
    out["rot_k"] = ks
    out["rot_out"] = np.stack([O.poly_mul_with_x_k(poly, int(kk)) for kk in ks])
    # decomposition (trgsw.rs:144-171) for (l, bgbit) = (3,6), (2,10), (1,22)
    trlwe = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
    trlwe[0, :4] = [0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF]
    out["dec_in"] = trlwe
    for l, bg in ((3, 6), (2, 10), (1, 22)):
        out[f"dec_out_{l}_{bg}"] = O.decomposition(trlwe, l, bg, O.gen_decomposition_offset(l, bg))
    # sample extract (trlwe.rs:106-136)
    out["se_in"] = trlwe
    out["se_k"] = np.array([0, 1, 700, 1023], np.int32)
    out["se_out"] = np.stack([O.sample_extract_index(trlwe, int(kk)) for kk in out["se_k"]])
    out["se2_out_n700"] = np.stack([O.sample_extract_index_2(trlwe, int(kk), 700) for kk in (0, 3, 699)])
    # gate prep (gates.rs:54-150), n = 16
    a = rng.integers(0, 2**32, 17, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, 17, dtype=np.uint64).astype(np.uint32)
    out["prep_a"], out["prep_b"] = a, b
    out["prep_out"] = np.stack([O.gate_prep(op, a, b, 16) for op in range(10)])
    # LUTs (lut/generator.rs:89-137)
    out["lut_id_m2"] = O.lut_generate(lambda x: x, 2)
    out["lut_not_m2"] = O.lut_generate(lambda x: 1 - x, 2)
    out["lut_id_m4"] = O.lut_generate(lambda x: x, 4)
    out["lut_sq_m16"] = O.lut_generate(lambda x: (x * x) % 16, 16)
    out["lut_id_m3"] = O.lut_generate(lambda x: x, 3)
    return out


def toy_bootstrap():
    """A complete tiny instance (n = 4) whose whole cloud key fits in a fixture."""
    P = O.Params("TOY_N4", 4, 3, 6, 2, 2, 2.0e-5, 2.0e-8)
    sk, ck = O.keygen(P, 77, with_time_domain=True)
    rng = np.random.default_rng(5)
    cts = rng.integers(0, 2**32, (6, P.n + 1), dtype=np.uint64).astype(np.uint32)
    cts[1, P.n] = 0            # b_tilda = 2N
    cts[2, P.n] = 0xFFFFFFFF   # b_tilda = 0 (non-wrapping usize add, trgsw.rs:202-203)
    cts[3, 0] = 0xFFFFFFFF     # a_tilda wraps to 0
    cts[4, :] = 0
    cts2 = rng.integers(0, 2**32, (6, P.n + 1), dtype=np.uint64).astype(np.uint32)
    out = {
        "params": np.array([P.n, P.l, P.bgbit, P.basebit, P.t], np.int32),
        "bsk": ck.bootstrapping_key,
        "bsk_time": ck.bootstrapping_key_time,
        "ksk": ck.key_switching_key,
        "offset": np.array([ck.decomposition_offset], np.uint32),
        "testvec": ck.blind_rotate_testvec,
        "cts": cts,
        "cts2": cts2,
    }
    br = O.batch_blind_rotate(ck, cts)
    out["blind_rotate"] = br
    out["blind_rotate_exact"] = np.stack([O.blind_rotate(ck, c, exact=True) for c in cts])
    out["lv1"] = np.stack([O.sample_extract_index(t, 0) for t in br])
    out["keyswitch"] = np.stack([O.identity_key_switching(ck, x) for x in out["lv1"]])
    out["bootstrap"] = O.batch_bootstrap(ck, cts)
    out["bootstrap_noks"] = O.batch_bootstrap(ck, cts, keyswitch=False)
    for op in range(10):
        out[f"gate_{op}"] = O.batch_gate(ck, op, cts, cts2)
    lut = O.lut_generate(lambda x: (3 * x + 1) % 4, 4)
    out["lut"] = lut
    out["bootstrap_lut"] = O.batch_bootstrap(ck, cts, testvec=lut)
    out["mux"] = O.batch_mux(ck, cts, cts2, cts[::-1].copy(), naive=False)
    out["mux_naive"] = O.batch_mux(ck, cts, cts2, cts[::-1].copy(), naive=True)
    idx = np.array([0, 1, 2, 3, 0, 2], np.int32)
    out["ep_index"] = idx
    out["ep_in"] = br
    out["ep_out"] = np.stack(
        [O.external_product_fft(ck.bootstrapping_key[i], t, P.l, P.bgbit, ck.decomposition_offset) for i, t in zip(idx, br)]
    )
    return out


if __name__ == "__main__":
    O.build()
    np.savez_compressed(os.path.join(HERE, "stage_vectors.npz"), **stage_vectors())
    np.savez_compressed(os.path.join(HERE, "toy_bootstrap.npz"), **toy_bootstrap())
    for f in ("stage_vectors.npz", "toy_bootstrap.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
