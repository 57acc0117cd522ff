#!/usr/bin/env python3
"""Headline benchmark: gate-bootstraps/sec (hom_nand, SECURITY_128_BIT) on N MI355X.

One "step" = one pass of the hot path (fused gate prep + blind rotate, then
sample-extract + identity key switch) over one batch of synthetic ciphertexts
per GPU, inputs and keys already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: 65,536 independent hom_nand bootstraps at
SECURITY_128_BIT; for N>1 every rank runs the same per-GPU batch on its own
shard (weak scaling, no data-path collective: the path shards embarrassingly).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the driver contract), with two extra objects:
  roofline     -- the dominant kernel (k_blind_rotate) against the resource that binds it: FP64 vector
                  arithmetic (f64 flops of the CMUX loop, counted from the ISA of the build being timed,
                  over the HIP-event launch duration, vs the 78.6 TFLOP/s FP64 vector peak), with the
                  shader clock and board power sampled during the timed steps, the VALU issue fraction
                  at that clock, the algorithmic-HBM figure of SURVEY 8(d) and the physical HBM fraction
  cpu_baseline -- the oracle (C port of the reference path, OpenMP over ciphertexts =
                  Rayon par_iter) on this box's host cores, thread-count sweep, bounded sample (N=1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# what each gate computes on plaintext bits, as the reference's tests assert it (src/gates.rs:553-700;
# Gates::xnor is XOR there: DESIGN.md quirk Q8)
GATE_TRUTH = {
    "nand": lambda a, b: ~(a & b), "or": lambda a, b: a | b, "and": lambda a, b: a & b,
    "xor": lambda a, b: a ^ b, "xnor": lambda a, b: a ^ b, "nor": lambda a, b: ~(a | b),
    "and_ny": lambda a, b: ~a & b, "and_yn": lambda a, b: a & ~b,
    "or_ny": lambda a, b: ~a | b, "or_yn": lambda a, b: a | ~b, "copy": lambda a, b: a,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="ciphertexts per GPU per step")
    ap.add_argument("--params", default="SECURITY_128_BIT")
    ap.add_argument("--gate", default="nand", help="gate name, or 'pbs' = LutBootstrap::bootstrap_lut (m=16, x^2 mod 16), "
                    "or 'mux' / 'mux_naive', or 'mixed' = half hom_mux + half hom_xor (BASELINE configs[4])")
    ap.add_argument("--modulus", type=int, default=16, help="message modulus of the 'pbs' workload (16 fits SECURITY_UINT4; "
                    "UINT1 / UINT2 / UINT3 take 2 / 4 / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool-devices", default=None, help="comma-separated device list: instead of the contract run, time "
                    "ONE process driving these devices through tfhe_hip_pool_batch_gate with host buffers (what a Rust "
                    "caller of the pool gets, PCIe included); prints its own JSON line")
    ap.add_argument("--pinned", action="store_true", help="with --pool-devices: operands in pinned host memory "
                    "(tfhe_hip_host_alloc): read and written in place over PCIe, no staging copies")
    ap.add_argument("--resident", action="store_true", help="with --pool-devices: the batch is RESIDENT on the first device's "
                    "GPU and goes through tfhe_hip_pool_batch_*_dev (shards scattered / gathered between the members by grouped "
                    "RCCL send / receive, or peer copies when a device repeats); prints scatter_ms / gather_ms")
    ap.add_argument("--stage", default=None, choices=["blind_rotate", "ifft", "fft", "poly_mul", "reencrypt"],
                    help="time one stage entry point instead of the gate path: the reference's criterion groups "
                    "`bootstrapping` (= trgsw::blind_rotate) and `fft_operations` (benches/gate_benchmarks.rs:77-125)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target wall time of the CPU sample (whole thread sweep)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the short runs of BASELINE configs[3] / [4] and "
                    "of the host-buffer path that the line carries as `other_configs`")
    ap.add_argument("--no-pool-resident", action="store_true", help="N > 1: skip the single-process run of BASELINE configs[2] "
                    "through ONE pool handle (batch resident on GPU 0, shards moved by RCCL) that the line carries as `pool_resident`")
    ap.add_argument("--oracle-sample", type=int, default=0, help="with --pool-devices --resident: hold the first K results to the "
                    "CPU path word for word (under the key exported from member 0)")
    return ap.parse_args()


def stage_mode(args):
    """benches/gate_benchmarks.rs:77-125: `bootstrapping` (one trgsw::blind_rotate) and the `fft_operations` group
    (fft_forward_1024 = FFTProcessor::ifft, fft_inverse_1024 = ::fft, poly_mul_1024), each as ONE call (criterion's
    shape) and as a batch.  blind_rotate runs device-resident; the FFT stage entry points take host buffers, so their
    figures include PCIe both ways (they exist for parity tests: the hot path never leaves the fused kernels)."""
    import numpy as np
    import torch

    import rs_tfhe_amd as R

    P = R.params.PARAM_SETS[args.params]
    eng = R.Engine(P, 0)
    rng = np.random.default_rng(7)
    N = R.params.N
    res = {"metric": f"stage `{args.stage}` ({args.params})", "stage": args.stage, "steps": args.steps, "warmup": args.warmup}

    def timed(fn, reps):
        for _ in range(max(1, args.warmup)):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    if args.stage == "blind_rotate":
        sk = R.SecretKey.new(P, seed=2024)
        eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
        B = args.batch
        cts = sk.encrypt_bool(rng.integers(0, 2, B).astype(bool), seed=5)
        tin = torch.from_numpy(cts.view(np.int32)).to("cuda:0")
        tout = torch.empty((B, 2, N), dtype=torch.int32, device="cuda:0")
        one = timed(lambda: eng.batch_blind_rotate_dev(tin[:1], tout[:1]), max(5, args.steps))
        eng.set_profiling(True)
        whole = timed(lambda: eng.batch_blind_rotate_dev(tin, tout), args.steps)
        kt = eng.kernel_times()
        res.update({"reference_bench": "bootstrapping (trgsw::blind_rotate), benches/gate_benchmarks.rs:77-90",
                    "single_call_ms": round(one * 1e3, 3), "batch": B, "batch_ms": round(whole * 1e3, 2),
                    "value": round(B / whole, 1), "unit": "blind rotations/s",
                    "kernel_ms_per_launch": round(kt["blind_rotate_ms"] / max(1, kt["blind_rotate_launches"]), 3),
                    "dispatch": eng.describe_dispatch(B), "device_resident": True})
    elif args.stage == "reencrypt":
        # proxy_reenc::reencrypt_tlwe_lv0 (src/proxy_reenc.rs:468-510; the reference has no criterion bench for it): one
        # ciphertext, and a device-resident batch -- one launch of the key-switch kernels (+ the padding kernel)
        from rs_tfhe_amd import proxy_reenc as PR

        alice, bob = R.SecretKey.new(P, seed=2024), R.SecretKey.new(P, seed=2026)
        rk = PR.ProxyReencryptionKey.new_symmetric(alice, bob, seed=2027)
        eng.load_reenc_key(rk.key_encryptions)
        B = args.batch
        bits = rng.integers(0, 2, B).astype(bool)
        cts = alice.encrypt_bool(bits, seed=5)
        tin = torch.from_numpy(cts.view(np.int32)).to("cuda:0")
        tout = torch.empty_like(tin)
        one = timed(lambda: eng.batch_reencrypt(cts[:1]), max(20, args.steps))
        eng.set_profiling(True)
        whole = timed(lambda: eng.batch_reencrypt_dev(tin, tout), args.steps)
        kt = eng.kernel_times()
        ok = bool(np.array_equal(bob.decrypt_bool(tout.cpu().numpy().view(np.uint32)), bits))
        res.update({"reference_bench": "proxy_reenc::reencrypt_tlwe_lv0, src/proxy_reenc.rs:468-510 (no criterion bench in the reference)",
                    "single_call_ms": round(one * 1e3, 4), "batch": B, "batch_ms": round(whole * 1e3, 3),
                    "value": round(B / whole, 1), "unit": "re-encryptions/s",
                    "kernel_ms_per_launch": round(kt["key_switch_ms"] / max(1, kt["key_switch_launches"]), 3),
                    "dispatch": eng.describe_dispatch(B).split(" ", 1)[1], "device_resident": True, "decrypt_ok": ok})
    else:
        B = min(args.batch, 16384)
        polys = rng.integers(0, 2**32, (B, N), dtype=np.uint64).astype(np.uint32)
        if args.stage == "ifft":
            fn1, fnB, name = (lambda: eng.batch_ifft(polys[:1])), (lambda: eng.batch_ifft(polys)), "fft_forward_1024 (FFTProcessor::ifft)"
        elif args.stage == "fft":
            spec = eng.batch_ifft(polys)
            fn1, fnB, name = (lambda: eng.batch_fft(spec[:1])), (lambda: eng.batch_fft(spec)), "fft_inverse_1024 (FFTProcessor::fft)"
        else:
            fn1, fnB, name = (lambda: eng.batch_poly_mul(polys[:1], polys[:1])), (lambda: eng.batch_poly_mul(polys, polys)), "poly_mul_1024"
        one = timed(fn1, max(20, args.steps))
        whole = timed(fnB, args.steps)
        res.update({"reference_bench": f"fft_operations/{name}, benches/gate_benchmarks.rs:92-125",
                    "single_call_ms": round(one * 1e3, 4), "batch": B, "batch_ms": round(whole * 1e3, 2),
                    "value": round(B / whole, 1), "unit": "polynomials/s", "pcie_inclusive": True})
    eng.close()
    print(json.dumps(res), flush=True)


def pool_mode(args):
    """One process, several devices, host buffers: the Rust caller's view of tfhe_hip_pool_*."""
    import numpy as np

    import rs_tfhe_amd as R

    devices = [int(d) for d in args.pool_devices.split(",")]
    P = R.params.PARAM_SETS[args.params]
    sk = R.SecretKey.new(P, seed=2024)
    pool = R.Pool(P, devices)
    t0 = time.perf_counter()
    pool.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    keygen_s = time.perf_counter() - t0
    B = args.batch * len(devices)
    if args.resident:
        return pool_resident_mode(args, R, pool, sk, devices, B, keygen_s)
    gate = R.engine.GATE_IDS[args.gate]
    rng = np.random.default_rng(1000)
    bits_a, bits_b = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    ca, cb = sk.encrypt_bool(bits_a, seed=11), sk.encrypt_bool(bits_b, seed=12)
    out = None
    if args.pinned:
        ca, cb, out = R.engine.pinned_copy(ca), R.engine.pinned_copy(cb), R.engine.pinned_empty(ca.shape)
    for _ in range(args.warmup):
        out = pool.batch_gate(gate, ca, cb, out=out)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pool.batch_gate(gate, ca, cb, out=out)
    elapsed = time.perf_counter() - t0
    ok = bool(np.array_equal(sk.decrypt_bool(out), GATE_TRUTH[args.gate](bits_a, bits_b)))
    print(json.dumps({
        "metric": f"gate-bootstraps/sec (hom_{args.gate}, {args.params}), single process, tfhe_hip_pool over host buffers",
        "value": round(B * args.steps / elapsed, 1), "unit": "bootstraps/s", "devices": devices, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "batch_total": B,
        "pcie_inclusive": True, "host_memory": "pinned (zero-copy)" if args.pinned else "pageable (staged)",
        "keygen_and_replication_s": round(keygen_s, 3), "decrypt_ok": ok}), flush=True)


def pool_resident_mode(args, R, pool, sk, devices, B, keygen_s):
    """The whole batch lives on the first device's GPU; tfhe_hip_pool_batch_*_dev cuts it over the members (shard r on
    member r), moves the shards device to device and brings the results back in input order.  `--gate mixed` is
    BASELINE configs[4]'s circuit level (half Gates::mux in the reference's formula, half hom_xor: two blind-rotation
    launches + one key switch) through the ONE pool handle."""
    import numpy as np
    import torch

    P = pool.params
    dev = torch.device("cuda", devices[0])
    rng = np.random.default_rng(1000)
    bits = [rng.integers(0, 2, B).astype(bool) for _ in range(3)]
    ta, tb, tc = (torch.from_numpy(sk.encrypt_bool(b, seed=11 + i).view(np.int32)).to(dev) for i, b in enumerate(bits))
    to = torch.empty_like(ta)
    h = B // 2
    codes = torch.full((B - h,), R.engine.XOR, dtype=torch.uint8, device=dev)
    gate = None if args.gate == "mixed" else R.engine.GATE_IDS[args.gate]

    def step():
        if args.gate == "mixed":
            mo, xo = R.circuit.mux_and_gates_dev(pool, ta[:h], tb[:h], tc[:h], codes, ta[h:], tb[h:])
            to[:h].copy_(mo)
            to[h:].copy_(xo)
        else:
            pool.batch_gate_dev(gate, ta, tb, to)

    def fence():
        pool.synchronize()
        torch.cuda.synchronize()

    with torch.cuda.device(dev):
        for _ in range(args.warmup):
            step()
        fence()
        pool.set_profiling(True)
        pool.transfer_times()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        tt = pool.transfer_times()
        pool.set_profiling(False)
    out = to.cpu().numpy().view(np.uint32)
    if args.gate == "mixed":
        ok = bool(np.array_equal(sk.decrypt_bool(out[h:]), bits[0][h:] ^ bits[1][h:]))
        boots = 2 * B
    else:
        ok = bool(np.array_equal(sk.decrypt_bool(out), GATE_TRUTH[args.gate](bits[0], bits[1])))
        boots = B
    calls = max(1, tt["calls"])
    oracle_equal, oracle_n = None, 0
    if args.oracle_sample > 0 and args.gate != "mixed":
        # the CPU path under the SAME key (exported from member 0), on the first K ciphertexts: checker only
        from oracle import oracle as O

        xk = pool.export_cloud_key(0)
        ock = O.CloudKey.from_arrays(O.PARAM_SETS[args.params], xk.bootstrapping_key, xk.key_switching_key,
                                     xk.decomposition_offset, xk.blind_rotate_testvec)
        oracle_n = min(args.oracle_sample, B)
        ha, hb = (t[:oracle_n].cpu().numpy().view(np.uint32) for t in (ta, tb))
        oracle_equal = bool(np.array_equal(O.batch_gate(ock, gate, ha, hb), out[:oracle_n]))
    print(json.dumps({
        "metric": f"gate-bootstraps/sec ({args.gate}, {args.params}), single process, tfhe_hip_pool_*_dev over a batch resident on device {devices[0]}",
        "value": round(boots * args.steps / elapsed, 1), "unit": "bootstraps/s", "devices": devices, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "batch_total": B,
        "resident_on": devices[0], "transport": pool.data_transport, "key_transport": pool.key_transport,
        # per pool call: the longest single shard transfer (transfers to different members overlap) and the sum over members
        "scatter_ms": round(tt["scatter_ms_max"], 3), "gather_ms": round(tt["gather_ms_max"], 3),
        # sums over the members' own brackets (under RCCL between distinct devices each is bounded only by its call's group
        # time: an upper bound, not a per-link cost), and the home stream's bracket around each call's whole group
        "scatter_ms_sum_per_call": round(tt["scatter_ms_sum"] / calls, 3), "gather_ms_sum_per_call": round(tt["gather_ms_sum"] / calls, 3),
        "scatter_group_ms": round(tt["scatter_group_ms_sum"] / calls, 3), "gather_group_ms": round(tt["gather_group_ms_sum"] / calls, 3),
        "scatter_MB_per_call": round(tt["scatter_bytes"] / calls / 1e6, 1), "gather_MB_per_call": round(tt["gather_bytes"] / calls / 1e6, 1),
        "pool_calls": tt["calls"],
        # set-up, each on its own (host wall clock): key generation on member 0, the communicator's creation
        # (ncclCommInitAll; 0 = the pool has none), the key's replication to the other members
        "keygen_enqueue_s": round(max(0.0, keygen_s - (tt["comm_create_ms"] + tt["key_replication_ms"]) * 1e-3), 3),
        "comm_create_s": round(tt["comm_create_ms"] * 1e-3, 3), "key_replication_s": round(tt["key_replication_ms"] * 1e-3, 3),
        # members that share a device exchange their shards by on-device copies: the times above then say nothing about
        # xGMI (a pool of distinct devices is what they are for)
        "transfers_cross_devices": len(set(devices)) == len(devices) and len(devices) > 1,
        "oracle_sample_equal": oracle_equal, "oracle_sample": oracle_n,
        "decrypt_ok": ok}), flush=True)


def key_switch_roofline(P, per_launch, ks_ms, ks_clk, pm, batch, plan=""):
    """Roofline of the second kernel.  Base-4 sets at batch sizes the matrix-core kernel takes: int8 MFMA ops of the
    one-hot contraction (2 x ciphertexts x 4*N*t rows x padded output columns x 4 byte planes) against the guide's
    measured i8 ceiling (3,944 TOPS, v_mfma_i32_16x16x64_i8; 32x32x32: 4,404), with the clock the kernel sustained
    (the matrix pipes are current-limited: 1.6-2.4 GHz depending on operand toggling,
    profiles/exp/logs/r3d_ubench_mfma.log).  Wider bases: the column-sliced kernel against the CUs' LDS read bandwidth.
    Anything else (forced fallback kernels): instruction-issue bound."""
    mfma = "key_switch=mfma" in plan  # what the library says it launched (tfhe_hip_describe_dispatch)
    out = {"avg_launch_ms": round(ks_ms, 3), "kernel": "k_key_switch_" + plan.split("key_switch=")[-1].split("(")[0] if plan else None}
    if mfma and ks_ms > 0:
        cols = -(-(P.n + 1) // 32) * 32
        ops = 2.0 * per_launch * (4 * 1024 * P.iks_t) * cols * 4
        tops = ops / (ks_ms * 1e-3) / 1e12
        mhz = ks_clk.get("shader_mhz") or None
        out.update({
            "kernel": "k_key_switch_mfma", "bound": "mfma_i8", "achieved": round(tops, 1), "peak": 3944.0, "unit": "TOP/s",
            "frac": round(tops / 3944.0, 4), "int8_ops_per_launch": ops, "shader_mhz": round(mhz, 1) if mhz else None,
            # the same ops against what the pipes deliver at the clock they were allowed: 1,024 SIMDs x 2,048 ops/clk
            "frac_at_sustained_clock": round(tops * 1e12 / (1024 * 2048 * mhz * 1e6), 4) if mhz else None,
        })
    elif "key_switch=sliced" in plan and ks_ms > 0:
        # column-sliced kernel (bases 16 .. 128): every ciphertext reads one 256-byte row slice per (group, 64-column
        # slice) from the LDS ring -- N*t groups x ceil((n+1)/64) slices -- which is what binds it since the round-4
        # rewrite (profiles/exp/logs/r4_ks_sl_ablation.log); peak: ds_read_b128 at 256 B/clk per CU, 256 CUs, 2.4 GHz
        lds_bytes = float(per_launch) * 1024 * P.iks_t * (-(-(P.n + 1) // 64)) * 256
        tbps = lds_bytes / (ks_ms * 1e-3) / 1e12
        peak = 256 * 256 * 2.4e9 / 1e12
        out.update({"bound": "lds_read", "achieved": round(tbps, 1), "peak": round(peak, 1), "unit": "TB/s (LDS)",
                    "frac": round(tbps / peak, 4), "lds_bytes_per_launch": lds_bytes,
                    "issue_frac": pm.get("key_switch", {}).get("issue_frac")})
    else:
        out.update({"bound": "valu+salu issue", "issue_frac": pm.get("key_switch", {}).get("issue_frac")})
    out["algorithmic_hbm_GBps"] = round(P.ksk_touched_bytes * per_launch / (ks_ms * 1e-3) / 1e9, 1) if ks_ms > 0 else None
    out["physical_hbm_frac"] = (round(pm["key_switch"]["hbm_bytes_per_launch"] / (ks_ms * 1e-3) / 8e12, 4)
                                if pm.get("key_switch", {}).get("hbm_bytes_per_launch") and ks_ms > 0 else None)
    return out


def single_gate_latency(eng, gate, ca, cb, schedule=((0.0, 60), (0.010, 50), (1.0, 5), (10.0, 1))):
    """BASELINE configs[0] through the GPU path: ONE `Gates::nand`-shaped call (host buffers in and out: two pageable
    H2D copies, the latency kernels, one D2H copy, one stream synchronise), which is what the reference's criterion
    `gate_nand` times on the CPU (benches/gate_benchmarks.rs:12-20).  A caller's gates do not arrive back to back, and
    an idle GPU drops its clocks, so the call is timed after idle gaps: `schedule` = (gap seconds, calls); each call
    sleeps `gap`, then runs and is timed on its own.  Returns, per gap, the median / min / max wall time of a call and
    the median time of its kernels (HIP events on the launch stream): wall - kernels = copies + launches + wake-up,
    and a kernel time that grows with the gap is the clock ramp."""
    import statistics

    # the very first call (in bench.py: right after the CPU baseline's ~20 s, GPU idle, the OpenMP team just released)
    eng.kernel_times()
    eng.set_profiling(True)
    t1 = time.perf_counter()
    eng.batch_gate(gate, ca[:1], cb[:1])
    first = (time.perf_counter() - t1) * 1e3
    kt = eng.kernel_times()
    eng.set_profiling(False)
    out = {"first_call": {"calls": 1, "wall_ms": round(first, 3), "kernels_ms": round(kt["blind_rotate_ms"] + kt["key_switch_ms"], 3)}}
    k = 0
    for gap, reps in schedule:
        wall, kern = [], []
        for _ in range(reps):
            i = k % len(ca)
            k += 1
            if gap:
                time.sleep(gap)
            t1 = time.perf_counter()
            eng.batch_gate(gate, ca[i:i + 1], cb[i:i + 1])
            wall.append((time.perf_counter() - t1) * 1e3)
        eng.kernel_times()
        eng.set_profiling(True)
        for _ in range(min(reps, 8 if gap < 1.0 else 3)):  # (the kernel-time pass sleeps too: keep the whole probe near 30 s)
            i = k % len(ca)
            k += 1
            if gap:
                time.sleep(gap)
            eng.batch_gate(gate, ca[i:i + 1], cb[i:i + 1])
            kt = eng.kernel_times()
            kern.append(kt["blind_rotate_ms"] + kt["key_switch_ms"])
        eng.set_profiling(False)
        out[f"{gap:g}s"] = {"calls": reps, "wall_ms_median": round(statistics.median(wall), 3), "wall_ms_min": round(min(wall), 3),
                            "wall_ms_max": round(max(wall), 3), "kernels_ms_median": round(statistics.median(kern), 3)}
    return out


def concurrent_single_gates(target, gate, ca, cb, threads=(1, 8, 64, 256), seconds=0.5):
    """The reference's strategies are `Send + Sync` (src/bootstrap/mod.rs:23): T host threads (C++,
    rs-tfhe_amd/csrc/callers.cpp -- a Rayon team's stand-in) each make one-ciphertext `Gates::nand`-shaped calls on ONE
    handle, back to back.  The library merges the calls in flight into shared launches (combine.hpp); the aggregate
    rate per team size, and the same calls with merging switched off (every call alone on the GPU, one after the
    other) for comparison."""
    import numpy as np

    from rs_tfhe_amd import callers

    rows, rates, med = len(ca), [], []
    for T in threads:
        est = 2.4e-3 * max(1.0, T / 256.0)
        K = max(3, min(rows // T, int(seconds / est)))
        n = T * K
        codes = np.full(n, gate, np.uint8)
        w = T * min(K, 2)
        callers.run(target, callers.OP_GATE, ca[:w], cb[:w], gates=codes[:w], threads=T, calls=min(K, 2))  # lanes, arenas, threads
        _, secs, ms = callers.run(target, callers.OP_GATE, ca[:n], cb[:n], gates=codes, threads=T, calls=K)
        rates.append(round(n / secs, 1))
        med.append(round(float(np.median(ms)), 3))
    bound = target.combine_stats()["max_count"]
    target.set_combining(0)
    try:
        T, K = 8, 12
        _, secs, _ = callers.run(target, callers.OP_GATE, ca[:T * K], cb[:T * K], gates=np.full(T * K, gate, np.uint8), threads=T, calls=K)
    finally:
        target.set_combining(bound)
    return {"threads": list(threads), "gates_per_s": rates, "call_ms_median": med,
            "unmerged_gates_per_s_8_threads": round(T * K / secs, 1),
            "what": "T host threads, one-ciphertext hom_nand calls with host buffers on one context, back to back"}


def child_line(argv, env_extra=None, timeout=300):
    """bench.py again as a FRESH child process (never an exec of this one: it has touched the GPU) with a hard timeout;
    the one JSON line it prints, or {"error": ...}.  The launcher's rendezvous variables are not passed on."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
                        "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS", "BENCH_SHARE_GPU")
           and not k.startswith("TORCHELASTIC_")}
    env.update(env_extra or {})
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(a) for a in argv], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout} s", "argv": [str(a) for a in argv]}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return {"error": f"rc {p.returncode}", "argv": [str(a) for a in argv], "stderr_tail": p.stderr[-600:]}
    return json.loads(lines[0])


def other_configs(args):
    """What BASELINE.json names beside configs[1], as short runs of their own (fresh child processes, three timed steps
    each): configs[3] (LutBootstrap::bootstrap_lut, SECURITY_UINT4, m = 16), the single-GPU share of configs[4] (half
    Gates::mux + half hom_xor, SECURITY_80_BIT), and the reference's own call shape -- host slices in, Vec out
    (src/gates.rs:352-383) -- through the pool handle with pageable and with pinned buffers."""
    out = {}
    for name, argv in (("configs3_pbs_uint4", ["--gate", "pbs", "--params", "SECURITY_UINT4"]),
                       ("configs4_share_mixed_80bit", ["--gate", "mixed", "--params", "SECURITY_80_BIT"])):
        d = child_line(["--gpus", 1, "--steps", 3, "--warmup", 1, "--batch", args.batch, "--no-cpu-baseline", "--no-other-configs"] + argv)
        out[name] = d if "error" in d else {
            "metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "decrypt_ok": d["decrypt_ok"], "roofline_frac": d["roofline"]["frac"], "roofline_kernel": d["roofline"]["kernel"],
            "avg_launch_ms": d["roofline"]["avg_launch_ms"], "key_switch_avg_launch_ms": d["roofline"]["key_switch_avg_launch_ms"],
            "dispatch": d["roofline"]["dispatch"], "workload": d["config"]["workload"]}
    for name, extra in (("host_path_pageable", []), ("host_path_pinned", ["--pinned"])):
        d = child_line(["--pool-devices", 0, "--steps", 2, "--warmup", 1, "--batch", args.batch, "--params", args.params,
                        "--gate", args.gate] + extra)
        out[name] = d if "error" in d else {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "host_memory", "pcie_inclusive", "decrypt_ok")}
    return out


def free_port() -> int:
    """A TCP port on 127.0.0.1 that is free now (bind to 0, read it back, release)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sock.bind(("127.0.0.1", 0))
        return int(sock.getsockname()[1])


def main():
    args = parse()
    if args.pool_devices:
        return pool_mode(args)
    if args.stage:
        return stage_mode(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not launched by torchrun: start it as a child (never exec after touching the GPU)
        # the rendezvous port: MASTER_PORT if the caller set one, otherwise a port the kernel says is free right now
        # (the driver's 1 -> 8 sweep starts this four times in a row on one node: a fixed port can meet a lingering
        # listener of the previous run)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT") or str(free_port()),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ncores = os.cpu_count() or 1
    if world > 1:
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, ncores // world)))

    import numpy as np
    import torch
    import torch.distributed as dist

    import rs_tfhe_amd as R  # the product: keys, ciphertexts and the timed path all come from it

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # BENCH_SHARE_GPU=1 is a plumbing self-test for boxes with fewer GPUs than ranks: ranks share
    # devices (rank % device_count) and rendezvous over gloo, since RCCL refuses two ranks on one GPU.
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    P = R.params.PARAM_SETS[args.params]
    special = args.gate in ("pbs", "mux", "mux_naive", "mixed")
    gate = None if special else R.engine.GATE_IDS[args.gate]
    B = args.batch

    # ---- synthetic, seeded inputs (one key for the whole job; shards differ by seed) ----
    # secret key: uniform bits (key.rs:39-46); cloud key: generated on rank 0's GPU and broadcast to the
    # other ranks' GPUs; ciphertexts: fresh encryptions of uniform bits / messages
    t0 = time.time()
    sk = R.SecretKey.new(P, seed=2024)  # explicit seeds: a benchmark wants reproducible inputs (never a real key)
    # the handle a Rust caller binds (tfhe_hip_pool_*): one member per rank here, because the
    # driver's contract is one process per GPU; the timed calls go to the member context's *_dev entry points
    pool = R.Pool(P, [local_rank])
    eng = R.Engine.from_pool(pool, 0)
    tk = time.perf_counter()
    if world == 1 or rank == 0:
        pool.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
        eng.synchronize()
    keygen_s = time.perf_counter() - tk
    key_broadcast_s = None
    if world > 1:  # ONE key, generated on rank 0 and replicated device to device (RCCL broadcast, engine layouts)
        dist.barrier()  # (the ranks that did not generate the key would otherwise count rank 0's key generation)
        tk = time.perf_counter()
        R.distributed.broadcast_engine_key(eng, src=0)
        torch.cuda.synchronize()
        key_broadcast_s = time.perf_counter() - tk
    rng = np.random.default_rng(1000 + rank)
    bits_a = rng.integers(0, 2, B).astype(bool)
    bits_b = rng.integers(0, 2, B).astype(bool)
    bits_c = rng.integers(0, 2, B).astype(bool)
    msgs = rng.integers(0, args.modulus, B)
    if args.gate == "pbs":
        ca = sk.encrypt_lwe_message(msgs, args.modulus, seed=11 + 3 * rank)
        cb = ca
    else:
        ca = sk.encrypt_bool(bits_a, seed=11 + 3 * rank)
        cb = sk.encrypt_bool(bits_b, seed=12 + 3 * rank)
    cc = sk.encrypt_bool(bits_c, seed=13 + 3 * rank) if args.gate.startswith("mux") or args.gate == "mixed" else None
    lut = R.lut.Generator(args.modulus).generate_lookup_table(lambda x: (x * x) % args.modulus) if args.gate == "pbs" else None
    setup_s = time.time() - t0

    ta = torch.from_numpy(ca.view(np.int32)).to(dev)
    tb = torch.from_numpy(cb.view(np.int32)).to(dev)
    to = torch.empty_like(ta)
    tc = torch.from_numpy(cc.view(np.int32)).to(dev) if cc is not None else None
    tlut = torch.from_numpy(lut.poly.view(np.int32)).to(dev) if lut is not None else None

    xor_codes = torch.full((B - B // 2,), R.engine.XOR, dtype=torch.uint8, device=dev) if args.gate == "mixed" else None

    def step():
        if args.gate == "pbs":
            eng.batch_bootstrap_dev(ta, to, testvec=tlut)
        elif args.gate.startswith("mux"):
            eng.batch_mux_dev(ta, tb, tc, to, naive=(args.gate == "mux_naive"))
        elif args.gate == "mixed":  # BASELINE configs[4]: half hom_mux (reference formula), half hom_xor
            h = B // 2                # one circuit level: two blind-rotation launches + one key switch in all
            mo, xo = R.circuit.mux_and_gates_dev(eng, ta[:h], tb[:h], tc[:h], xor_codes, ta[h:], tb[h:])
            to[:h].copy_(mo)
            to[h:].copy_(xo)
        else:
            eng.batch_gate_dev(gate, ta, tb, to)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.kernel_times()  # reset
    eng.clock_sample()
    eng.set_profiling(True)  # HIP events around each kernel, on the launch stream; shader-clock sampling in the kernel
    # board power while the timed steps run: hwmon power1_input of the busiest GPU of the box (this rank's, at N=1)
    import glob
    import threading

    pfiles = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") if rank == 0 else []
    watts, stop = [], threading.Event()

    def sample_power():
        while not stop.is_set():
            best = 0.0
            for f in pfiles:
                try:
                    best = max(best, float(open(f).read()) * 1e-6)
                except (OSError, ValueError):
                    pass
            watts.append(best)
            time.sleep(0.02)

    sampler = threading.Thread(target=sample_power, daemon=True)
    if pfiles:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stop.set()
    if pfiles:
        sampler.join()
    eng.set_profiling(False)
    kt = eng.kernel_times()
    clk = eng.clock_sample()
    ks_clk = eng.key_switch_clock_sample()
    watts = sorted(watts[len(watts) // 4:])  # drop the ramp
    power_w = round(watts[len(watts) // 2]) if watts and watts[-1] > 0 else None
    power_cap_w = None
    try:
        caps = [float(open(f).read()) * 1e-6 for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")]
        power_cap_w = round(max(caps)) if caps else None
    except (OSError, ValueError):
        pass

    # per-rank record (so that a bad scaling curve can be read from the line alone): this rank's own wall time for
    # the K steps, its kernels' average launch times, its shader clock and its key-broadcast time
    per_rank = None
    if world > 1:
        mine = torch.tensor([elapsed, kt["blind_rotate_ms"] / max(1, kt["blind_rotate_launches"]),
                             kt["key_switch_ms"] / max(1, kt["key_switch_launches"]), clk["shader_mhz"] or 0.0,
                             key_broadcast_s or 0.0], dtype=torch.float64, device="cpu" if share else dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = [e.cpu().tolist() for e in every]
        per_rank = {
            "ms_per_step": [round(e[0] / args.steps * 1e3, 2) for e in every],
            "blind_rotate_ms": [round(e[1], 2) for e in every],
            "key_switch_ms": [round(e[2], 3) for e in every],
            "shader_mhz": [round(e[3]) for e in every],
            "key_broadcast_s": [round(e[4], 3) for e in every],
        }
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sanity: decrypt the shard (integer, host) ----
    out = to.cpu().numpy().view(np.uint32)
    if args.gate == "pbs":
        decrypt_ok = bool(np.array_equal(sk.decrypt_lwe_message(out, args.modulus), (msgs ** 2) % args.modulus))
    elif args.gate == "mux":
        decrypt_ok = None  # Gates::mux is the reference formula: no decrypt claim
    elif args.gate == "mixed":  # claim only the xor half (the mux half is the Q5 formula)
        h = B // 2
        decrypt_ok = bool(np.array_equal(sk.decrypt_bool(out[h:]), bits_a[h:] ^ bits_b[h:]))
    elif args.gate == "mux_naive":
        decrypt_ok = bool(np.array_equal(sk.decrypt_bool(out), np.where(bits_a, bits_b, bits_c)))
    else:
        decrypt_ok = bool(np.array_equal(sk.decrypt_bool(out), GATE_TRUTH[args.gate](bits_a, bits_b)))

    if world > 1 and decrypt_ok is not None:  # every rank checked its own shard: the line's flag is the AND over ranks
        flag = torch.tensor([1 if decrypt_ok else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        decrypt_ok = bool(flag.item())
    key_broadcast_backend = dist.get_backend() if world > 1 else None
    if rank != 0:
        if world > 1:
            pool.close()  # give the GPU back before rank 0 starts the pool-resident child
            if not args.no_pool_resident:
                dist.barrier()
            dist.destroy_process_group()
        return

    # mux and mux_naive count 3 bootstraps, a plain gate 1; "mixed" is half and half
    boots_per_item = 3 if args.gate.startswith("mux") else (2 if args.gate == "mixed" else 1)
    value = world * B * boots_per_item * args.steps / elapsed
    bytes_per_bootstrap = P.algorithmic_bytes_per_bootstrap(1 if args.gate == "pbs" else 2)
    # dominant kernel: k_blind_rotate.  Algorithmic bytes per launch = B * (BSK once +
    # two input TLWEs + the extracted level-1 TLWE it writes)  (DESIGN.md "Roofline accounting")
    br_bytes_per_ct = P.bsk_bytes + (1 if args.gate == "pbs" else 2) * P.tlwe_lv0_bytes + (R.params.N + 1) * 4
    br_ms = kt["blind_rotate_ms"] / max(1, kt["blind_rotate_launches"])
    ks_ms = kt["key_switch_ms"] / max(1, kt["key_switch_launches"])
    # launches may be fewer ciphertexts than B only if chunking is on; per-launch units = bootstraps / launches
    per_launch = kt["bootstraps"] / max(1, kt["blind_rotate_launches"])
    achieved = (br_bytes_per_ct * per_launch) / (br_ms * 1e-3) / 1e9 if br_ms > 0 else 0.0
    # PMC counters cannot be read live: per-launch figures for this exact workload from the committed profile
    # (profiles/pmc_roofline.json, written by profiles/collect.sh from separate --pmc passes)
    # An entry is quoted only if it was measured on THIS tree: its stamp (digest of rs-tfhe_amd/csrc/*) must equal
    # the digest of the sources being timed; otherwise `traffic` is null and traffic_source says why.
    import hashlib

    hsrc = hashlib.sha256()
    for name in ("blind_rotate.hpp", "blind_rotate_wide.hpp", "experiment.hpp", "fft512.hpp", "key_switch.hpp", "key_switch_mfma.hpp", "keygen.hpp"):
        hsrc.update(name.encode())  # device code only (the same list as profiles/pmc_roofline.py)
        hsrc.update(open(os.path.join(ROOT, "rs-tfhe_amd", "csrc", name), "rb").read())
    csrc_sha = hsrc.hexdigest()[:16]
    cur_plan = eng.describe_dispatch(B)
    pm, traffic_source = {}, "none: no profiles/pmc_roofline.json entry for this workload"
    try:
        for entry in json.load(open(os.path.join(ROOT, "profiles", "pmc_roofline.json"))):
            if entry["config"] == {"params": args.params, "batch": B, "gate": args.gate}:
                # same device code AND (entries that record it) the same dispatch plan: the rules that pick kernels and
                # grids are host code, outside the digest
                if entry.get("source", {}).get("csrc_sha256") == csrc_sha and entry.get("dispatch") in (None, cur_plan):
                    pm = entry
                    traffic_source = f"profiles/pmc_roofline.json entry '{entry['tag']}' (separate --pmc passes, same kernel sources {csrc_sha})"
                elif not pm:
                    traffic_source = (f"none: entry '{entry['tag']}' was measured on other kernel sources "
                                      f"({entry.get('source', {}).get('csrc_sha256', 'unstamped')} != {csrc_sha}); re-run profiles/collect.sh")
    except Exception:
        pass
    traffic = pm.get("blind_rotate", {}).get("hbm_bytes_per_launch")
    # Instruction mix of one CMUX step of the build being timed (rs-tfhe_amd/kernel_isa.json, written by `make`
    # from the compiler's gfx950 assembly; cross-check: SQ_INSTS_VALU per launch / (batch * n) in profiles/).
    try:
        isa = json.load(open(os.path.join(ROOT, "rs-tfhe_amd", "kernel_isa.json")))[f"l{P.l}"]
        isa_source = "rs-tfhe_amd/kernel_isa.json (this build)"
    except (OSError, KeyError, ValueError):
        # library built without the Makefile's ISA step: the executed-flop roofline cannot be priced for THIS build.
        # The ALGORITHMIC count of SURVEY 8(d) (258,048 flops per CMUX step at l = 3, scaled by the FFT count for
        # other l) is used instead and the line says so; the ISA-derived issue fractions are null.
        alg = ((2 * P.l + 2) * 26112 + 2 * P.l * 8192) // 64  # per lane: (2l+2) transforms + 2l x 2 x 512 complex MACs; 258,048 / 64 at l = 3
        isa = {"f64_flop_per_lane": alg, "valu": None, "f64_fma": None, "f64_add": None, "f64_mul": None, "f64_other": None}
        isa_source = "MISSING rs-tfhe_amd/kernel_isa.json: algorithmic flops only (run `make -C rs-tfhe_amd/csrc`)"
        print("bench.py: " + isa_source, file=sys.stderr)
    wave_steps_per_s = P.n * per_launch / (br_ms * 1e-3) if br_ms > 0 else 0.0  # CMUX steps of one wave, whole chip
    tflops = wave_steps_per_s * 64 * isa["f64_flop_per_lane"] / 1e12
    # SURVEY 8(d)'s ALGORITHMIC flops per CMUX step (the contract figure, `roofline.frac`): (2l+2) transforms of 26,112
    # flops + 2l digit rows x 2 spectra x 512 complex MACs of 8 flops = 258,048 at l = 3, 120,832 at l = 1.
    # `frac_executed` prices the flops the kernel EXECUTES (from its ISA: a few per cent more -- twiddle folding, the
    # rounding); both are printed.
    alg_flop_per_step = (2 * P.l + 2) * 26112 + 2 * P.l * 8192
    tflops_alg = wave_steps_per_s * alg_flop_per_step / 1e12
    plan = eng.describe_dispatch(int(per_launch)) if per_launch else ""
    shader_mhz = clk["shader_mhz"] or None
    # issue slots: 1,024 SIMDs, one FP64 wave-instruction per 4 cycles (16 lanes/clk), at the clock the kernel ran at
    have_isa = isa["valu"] is not None
    f64_instr = (isa["f64_fma"] + isa["f64_add"] + isa["f64_mul"] + isa.get("f64_other", 0)) if have_isa else None
    simd_cycles_per_s = 1024 * (shader_mhz or 2400.0) * 1e6
    roofline = {
        "kernel": f"k_blind_rotate<{P.l}>",
        "bound": "fp64_valu",
        # the contract figure: SURVEY 8(d)'s ALGORITHMIC flops per launch over the HIP-event launch duration
        "achieved": round(tflops_alg, 2),
        "peak": 78.6,
        "unit": "TFLOP/s",
        "frac": round(tflops_alg / 78.6, 4),
        "algorithmic_flop_per_cmux_step": alg_flop_per_step,
        # the flops the kernel EXECUTES (from the ISA of this build: a few per cent more -- twiddle folding, rounding)
        "achieved_executed": round(tflops, 2),
        "frac_executed": round(tflops / 78.6, 4),
        "dispatch": plan,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "avg_launch_ms": round(br_ms, 3),
        # under the board's power cap time tracks energy, not cycles: the quantity that moves
        "joules_per_launch": round(power_w * br_ms * 1e-3, 1) if power_w else None,
        "microjoules_per_bootstrap": round(power_w * br_ms * 1e-3 / per_launch * 1e6, 2) if power_w and per_launch else None,
        "f64_flop_per_lane_per_cmux_step": isa["f64_flop_per_lane"],
        "valu_instr_per_cmux_step": isa["valu"],
        "instruction_mix_source": isa_source,
        "shader_mhz": round(shader_mhz, 1) if shader_mhz else None,
        "board_power_w": power_w,
        "board_power_cap_w": power_cap_w,
        # the same flops against the FP64 peak AT THE SUSTAINED CLOCK (the board is power-capped: DESIGN.md section 5)
        "frac_at_sustained_clock": round(tflops_alg / (78.6 * shader_mhz / 2400.0), 4) if shader_mhz else None,
        "frac_executed_at_sustained_clock": round(tflops / (78.6 * shader_mhz / 2400.0), 4) if shader_mhz else None,
        # what this board sustains on nothing but v_fma_f64 over random operands (profiles/exp/logs/r2u_ubench_random_operands.log):
        # the 1,400 W cap holds that stream at 2,027 MHz = 66.4 TFLOP/s
        "frac_executed_of_power_capped_fma_peak": round(tflops / 66.4, 4),
        "valu_issue_frac": round(wave_steps_per_s * isa["valu"] * 4 / simd_cycles_per_s, 4) if have_isa else None,
        "f64_issue_frac": round(wave_steps_per_s * f64_instr * 4 / simd_cycles_per_s, 4) if have_isa else None,
        # every bootstrap "consumes" the whole key once.  The key is shared through L1/L2, so this
        # exceeds the HBM peak by construction and is NOT a roofline fraction; physical_hbm_frac is.
        "algorithmic_hbm": {
            "note": "figure of merit, NOT a bound: the key is shared by every ciphertext in flight (L1/L2 hits), so the ratio exceeds 1; physical_hbm_frac is the fraction of HBM bandwidth used",
            "achieved_GBps": round(achieved, 1), "peak_GBps": 8000.0, "ratio": round(achieved / 8000.0, 4),
            "bytes_per_launch": int(br_bytes_per_ct * per_launch),
            "whole_path_GBps": round(value / world * bytes_per_bootstrap / 1e9, 1),
        },
        "physical_hbm_frac": round(traffic / (br_ms * 1e-3) / 8e12, 4) if traffic and br_ms > 0 else None,
        # flat copies of what the sub-objects hold, for parsers that keep only scalars
        "algorithmic_hbm_GBps": round(achieved, 1),
        "algorithmic_hbm_ratio_to_peak": round(achieved / 8000.0, 4),
        "whole_path_algorithmic_GBps": round(value / world * bytes_per_bootstrap / 1e9, 1),
        "key_switch_avg_launch_ms": round(ks_ms, 3),
        "key_switch": key_switch_roofline(P, per_launch, ks_ms, ks_clk, pm, B, plan),
    }

    cpu = None
    if world == 1 and not args.no_cpu_baseline and not special:
        # The CPU leg: the oracle (C port of the reference path) under the SAME key, exported from the
        # engine, on the same ciphertexts.  This is the only place bench.py touches oracle/.
        from oracle import oracle as O

        xk = eng.export_cloud_key()
        ock = O.CloudKey.from_arrays(O.PARAM_SETS[args.params], xk.bootstrapping_key, xk.key_switching_key,
                                     xk.decomposition_offset, xk.blind_rotate_testvec)
        allt = O.num_threads()
        try:
            affinity = len(os.sched_getaffinity(0))
        except AttributeError:
            affinity = ncores
        try:
            cgroup_cpu_max = open("/sys/fs/cgroup/cpu.max").read().strip()
        except OSError:
            cgroup_cpu_max = None
        # Rayon's par_iter uses every logical CPU; sweep the team size and report the best rate, so that an
        # oversubscribed or bandwidth-starved full team does not stand for "the CPU" (each thread streams the whole
        # 68.8 MB bootstrapping key per bootstrap: the batch is ciphertext-major, as gates.rs:357-383 is)
        quota = None  # CPUs' worth of time the container may use ("max" = unlimited)
        try:
            q, per_ = cgroup_cpu_max.split()
            quota = float(q) / float(per_) if q != "max" else None
        except (AttributeError, ValueError):
            pass
        sweep = sorted({t for t in (1, 8, int(quota) if quota else 16, 32, 64, allt) if 1 <= t <= max(allt, 1)})
        budget = args.cpu_seconds / len(sweep)
        sweep_res, best, ref = [], None, None
        for t in sweep:
            t1 = time.perf_counter()
            O.batch_gate(ock, gate, ca[:t], cb[:t], nthreads=t)
            per = max(1e-3, time.perf_counter() - t1)
            cnt = int(min(B, max(t, t * int(budget / per))))
            t1 = time.perf_counter()
            r_ = O.batch_gate(ock, gate, ca[:cnt], cb[:cnt], nthreads=t)
            dt = time.perf_counter() - t1
            sweep_res.append({"threads": t, "bootstraps_per_s": round(cnt / dt, 2), "sample": cnt, "seconds": round(dt, 2)})
            if best is None or cnt / dt > best[0]:
                best = (cnt / dt, t, cnt, dt)
            if ref is None or len(r_) > len(ref):
                ref = r_
        rate1 = sweep_res[0]["bootstraps_per_s"] if sweep_res[0]["threads"] == 1 else None
        for e in sweep_res:
            e["parallel_efficiency"] = round(e["bootstraps_per_s"] / (rate1 * e["threads"]), 3) if rate1 else None
        threads, sample, cpu_s = best[1], best[2], best[3]
        # BASELINE configs[0]: one hom_nand gate on one core (criterion gate_nand, benches/gate_benchmarks.rs:12-20)
        singles = []
        for r_ in range(9):  # median of nine: the first calls after the sweep run on cold caches and a parked core
            t1 = time.perf_counter()
            O.batch_gate(ock, gate, ca[r_:r_ + 1], cb[r_:r_ + 1], nthreads=1)
            singles.append((time.perf_counter() - t1) * 1e3)
        single_ms = sorted(singles)[len(singles) // 2]
        O.batch_gate(ock, gate, ca[:1], cb[:1], nthreads=allt)  # restore the OpenMP team size
        # the same single gate through the GPU path (host buffers, includes PCIe + sync), back to back and after idle
        # gaps (single_gate_latency): `gpu_single_gate_ms_warm` = median of back-to-back calls, `..._after_idle` =
        # median of calls that each follow 1 s of idle; the whole table is in `gpu_single_gate_latency`
        lat = single_gate_latency(eng, gate, ca, cb)
        conc = concurrent_single_gates(eng, gate, ca, cb)
        cpu = {
            "value": round(best[0], 2),
            "single_gate_ms_1core": round(single_ms, 2),
            "gpu_single_gate_ms_warm": lat["0s"]["wall_ms_median"],
            "gpu_single_gate_ms_after_idle": lat["1s"]["wall_ms_median"],
            "gpu_single_gate_latency": lat,
            "gpu_concurrent_single_gate": conc,
            "unit": "bootstraps/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{sample} of the same {args.gate} batch ({args.params}), OpenMP over ciphertexts, {cpu_s:.1f} s "
                      f"(best of a thread sweep, {sum(e['seconds'] for e in sweep_res):.0f} s in all)",
            "thread_sweep": sweep_res,
            "host": {"os_cpu_count": ncores, "sched_getaffinity": affinity, "cgroup_cpu_max": cgroup_cpu_max,
                     "cpu_quota": quota, "omp_max_threads": allt},
            "gpu_matches_cpu_bit_exact": bool(np.array_equal(ref, out[:len(ref)])),
        }

    others = None
    if world == 1 and not args.no_other_configs and not special and args.params == "SECURITY_128_BIT":
        pool.close()  # (the children use this GPU: give the key and the staging back first)
        del ta, tb, to
        torch.cuda.empty_cache()
        others = other_configs(args)
    resident = None
    if world > 1 and not args.no_pool_resident:
        # BASELINE configs[2] as its text has it -- "sharded across 8 x MI355X via RCCL over xGMI": ONE caller, ONE pool
        # handle over all N devices, the GLOBAL batch resident on GPU 0, shards scattered / gathered by grouped
        # ncclSend / ncclRecv (tfhe_hip_pool_batch_gate_dev; src/gates.rs:357-383 is the call it replaces).  The ranks
        # have finished: they leave the process group and give their contexts back, then rank 0 alone starts a fresh
        # child with a hard timeout.  The headline above never depends on it.
        n_dev = torch.cuda.device_count()
        devs = ",".join(str(r % n_dev) for r in range(world)) if share else ",".join(str(r) for r in range(world))
        dist.barrier()
        dist.destroy_process_group()
        pool.close()
        del ta, tb, to
        torch.cuda.empty_cache()
        d = child_line(["--pool-devices", devs, "--resident", "--steps", min(args.steps, 3), "--warmup", 1, "--batch", B,
                        "--params", args.params, "--gate", args.gate if args.gate in ("mixed",) or not special else "nand",
                        "--oracle-sample", 16], {"TFHE_HIP_POOL_RCCL": "1"}, timeout=420)
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "batch_total", "devices", "transport", "key_transport", "scatter_ms",
                "gather_ms", "scatter_group_ms", "gather_group_ms", "scatter_MB_per_call", "gather_MB_per_call", "comm_create_s",
                "key_replication_s", "transfers_cross_devices", "decrypt_ok", "oracle_sample_equal", "oracle_sample")
        resident = d if "error" in d else {k: d.get(k) for k in keep}

    line = {
        "metric": f"gate-bootstraps/sec (hom_{args.gate}, {args.params})" if not special else
                  f"gate-bootstraps/sec ({args.gate}, {args.params})",
        "value": round(value, 1),
        "unit": "bootstraps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"batch of {B} independent hom_{args.gate} bootstraps per GPU, {args.params} (N=1024)",
            "batch_per_gpu": B,
            "global_batch": B * world,
            "parallelism": f"batch-sharded x{world}, one process per GPU, no data-path collective",
            "n": P.n, "l": P.l, "bgbit": P.bgbit, "basebit": P.basebit, "t": P.iks_t,
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "other_configs": others,
        "pool_resident": resident,
        "decrypt_ok": decrypt_ok,
        "setup_s": round(setup_s, 1),
        "keygen_s": round(keygen_s, 3),
        # N > 1: how the one key reached the other ranks' GPUs and how long that took (max over ranks); the data path
        # itself has no collective
        "key_broadcast_s": round(max(per_rank["key_broadcast_s"]), 3) if per_rank else None,
        "key_broadcast_backend": key_broadcast_backend,
        "per_rank": per_rank,
        "ms_per_step_min_rank": min(per_rank["ms_per_step"]) if per_rank else None,
        "ms_per_step_max_rank": max(per_rank["ms_per_step"]) if per_rank else None,
    }
    print(json.dumps(line), flush=True)
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
