"""Hardware-day tests: what only a box with SEVERAL MI355X can show -- shards and keys over physical
xGMI links by RCCL and by peer copies, the 1 -> N scaling of the contract bench, BASELINE configs[2] through one pool
handle over distinct devices.  Every test skips unless the box has at least two GPUs, and needs no edits on one that
does: the device list is range(device_count()).  profiles/hardware_day.sh runs this file beside the bench sweep.

(The code paths themselves -- staging, shard arithmetic, events, the RCCL symbols -- run on the one-GPU pool every
round: tests/test_gpu_pool_resident.py with devices = [0] * 8 and a self send / receive.  Setting
TFHE_HIP_TEST_DEVICES=0,0 runs THIS file's bodies on one GPU as a dry run, with the assertions that need distinct
devices left out.)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1024


def _devices():
    import torch

    env = os.environ.get("TFHE_HIP_TEST_DEVICES")
    if env:
        return [int(d) for d in env.split(",")]
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 GPUs (this box has {n})")
    return list(range(n))


def _distinct(devs):
    return len(set(devs)) == len(devs) and len(devs) > 1


def _dev(a, device):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(f"cuda:{device}")


def _host(t):
    return t.cpu().numpy().view(np.uint32)


def _bench(argv, env_extra=None, timeout=1800):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in argv], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("rccl", ["1", "0"])
def test_key_replication_and_every_entry_point_over_distinct_devices(O, keys128, monkeypatch, rccl):
    """The cloud key reaches every member by ncclBroadcast (TFHE_HIP_POOL_RCCL=1) and by hipMemcpyPeer (=0), identical on
    all of them; a batch resident on the first and on the last member's GPU is cut over all members, moved, bootstrapped
    and gathered in input order -- gates, mixed gates, a programmable bootstrap, mux -- equal to one context word for word."""
    import torch

    import rs_tfhe_amd as R
    from test_gpu_parity import _cloud_key

    devs = _devices()
    monkeypatch.setenv("TFHE_HIP_POOL_RCCL", rccl)
    sk, ck = keys128
    pk = _cloud_key(ck)
    pool = R.Pool(pk.params, devs)
    pool.load_cloud_key(pk)
    if _distinct(devs):
        assert pool.key_transport == ("rccl" if rccl == "1" else "peer-copy")
    k0 = pool.export_cloud_key(0)
    for m in range(1, len(devs)):
        km = pool.export_cloud_key(m)
        assert np.array_equal(km.bootstrapping_key, k0.bootstrapping_key) and np.array_equal(km.key_switching_key, k0.key_switching_key), m
    assert np.array_equal(k0.key_switching_key, pk.key_switching_key)
    single = R.Engine(pk.params, devs[0])
    single.load_cloud_key(pk)
    rng = np.random.default_rng(9500)
    count = 256 * len(devs) + 37
    A, B, Cc = (rng.integers(0, 2, count).astype(bool) for _ in range(3))
    ca, cb, cc = sk.encrypt_bool(A, 9501), sk.encrypt_bool(B, 9502), sk.encrypt_bool(Cc, 9503)
    codes = rng.integers(0, 11, count).astype(np.uint8)
    tv = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
    want = {"nand": single.batch_gate(O.GATE_NAND, ca, cb), "mixed": single.batch_gates_mixed(codes, ca, cb),
            "lut": single.batch_bootstrap(ca, tv), "mux": single.batch_mux(ca, cb, cc, naive=True)}
    assert np.array_equal(want["nand"][:32], O.batch_gate(ck, O.GATE_NAND, ca[:32], cb[:32]))
    for home in sorted({0, len(devs) - 1}):
        d = devs[home]
        with torch.cuda.device(d):
            ta, tb, tc = _dev(ca, d), _dev(cb, d), _dev(cc, d)
            tcodes, ttv = torch.from_numpy(codes).to(f"cuda:{d}"), _dev(tv, d)
            to = torch.zeros_like(ta)
            pool.batch_gate_dev(O.GATE_NAND, ta, tb, to, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(to), want["nand"]), (home, "nand")
            if _distinct(devs):
                assert pool.data_transport == ("rccl" if rccl == "1" else "peer-copy")
            pool.batch_gates_mixed_dev(tcodes, ta, tb, to, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(to), want["mixed"]), (home, "mixed")
            pool.batch_bootstrap_dev(ta, to, testvec=ttv, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(to), want["lut"]), (home, "lut")
            pool.batch_mux_dev(ta, tb, tc, to, naive=True, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(to), want["mux"]), (home, "mux")
    # host-pointer calls: large ones are cut over the members, small ones from a team of threads go to the least loaded
    assert np.array_equal(pool.batch_gate(O.GATE_NAND, ca, cb), want["nand"])
    from rs_tfhe_amd import callers

    T, K = 8 * len(devs), 6
    n = T * K
    pool.combine_stats()
    out, _, _ = callers.run(pool, callers.OP_GATE, ca[:n], cb[:n], gates=np.full(n, O.GATE_NAND, np.uint8), threads=T, calls=K)
    st = pool.combine_stats()
    assert np.array_equal(out, want["nand"][:n])
    assert sum(s["requests"] for s in st) == n, st
    if _distinct(devs):  # (members that share a device count once: a dry run on {0, 0} sends everything to the first)
        assert sum(1 for s in st if s["requests"] > 0) >= 2, st
    single.close()
    pool.close()


def test_configs2_through_one_pool_handle_over_every_gpu(O):
    """BASELINE configs[2]: 65,536 x N hom_nand resident on GPU 0, ONE pool handle over all N GPUs, shards by grouped
    ncclSend / ncclRecv over xGMI (bench.py --pool-devices 0..N-1 --resident), and the same with peer copies.  Expected
    on eight MI355X (367.5 MB in, 183.8 MB out per peer at 153 GB/s per link): scatter <= 10 ms per peer,
    gather <= 5 ms, the whole job within 15 % of N x the one-GPU rate."""
    devs = _devices()
    arg = ",".join(str(d) for d in devs)
    one = _bench(["--gpus", 1, "--steps", 3, "--warmup", 1, "--no-cpu-baseline", "--no-other-configs"])
    for rccl in ("1", "0"):
        d = _bench(["--pool-devices", arg, "--resident", "--steps", 3, "--warmup", 1, "--oracle-sample", 16], {"TFHE_HIP_POOL_RCCL": rccl})
        assert d["decrypt_ok"] is True and d["oracle_sample_equal"] is True and d["batch_total"] == 65536 * len(devs)
        if _distinct(devs):
            assert d["transport"] == ("rccl" if rccl == "1" else "peer-copy") and d["transfers_cross_devices"] is True
            assert d["key_transport"] == ("rccl" if rccl == "1" else "peer-copy")
            assert d["scatter_ms"] <= 10.0 and d["gather_ms"] <= 5.0, d
            assert d["value"] >= 0.85 * len(devs) * one["value"], (d["value"], one["value"])
        print(json.dumps({k: d[k] for k in ("devices", "transport", "value", "ms_per_step", "scatter_ms", "gather_ms", "scatter_group_ms",
                                            "gather_group_ms", "comm_create_s", "key_replication_s")}))


def test_contract_bench_scales_over_the_gpus():
    """`bench.py --gpus N` for N = 1, 2, 4, ... up to the box: one process per GPU over RCCL, the key broadcast, every rank
    its own shard, no data-path collective -- near-linear (>= 7.5 x at 8), and the N > 1 line carries the pool-resident
    run of the same global batch."""
    devs = _devices()
    if not _distinct(devs):
        pytest.skip("dry run on one GPU: the scaling curve needs distinct devices")
    sizes = [n for n in (1, 2, 4, 8) if n <= len(devs)]
    values = {}
    for n in sizes:
        d = _bench(["--gpus", n, "--steps", 5, "--warmup", 2, "--no-cpu-baseline", "--no-other-configs"])
        assert d["n_gpus"] == n and d["decrypt_ok"] is True
        values[n] = d["value"]
        if n > 1:
            assert d["key_broadcast_backend"] == "nccl" and d["key_broadcast_s"] > 0
            pr = d["pool_resident"]
            assert "error" not in pr, pr
            assert pr["transport"] == "rccl" and pr["decrypt_ok"] is True and pr["oracle_sample_equal"] is True
            assert values[n] >= 0.9375 * n * values[1], values  # 7.5 / 8
    print(json.dumps(values))
