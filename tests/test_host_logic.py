"""Host-side logic of the product package (no GPU, no compute calls): parameter data,
LUT construction vs the oracle, sharding arithmetic, and that the C-ABI library loads
and exports every symbol include/tfhe_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    declared = set(re.findall(r"\b(tfhe_hip_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    from rs_tfhe_amd import _capi

    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libtfhe_hip.so"
    assert _capi.lib().tfhe_hip_name() == b"hip-gfx950"


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/tfhe_hip.h must compile as C99 (no C++-isms), which is what a
    cgo / bindgen / ctypes consumer sees."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    src = tmp_path / "hdr.c"
    src.write_text('#include "tfhe_hip.h"\nint main(void) { tfhe_hip_pool *p = 0; tfhe_hip_ctx *c = 0; (void)p; (void)c; '
                   'return sizeof(tfhe_hip_kernel_times) + sizeof(tfhe_hip_clock_sample) + sizeof(tfhe_hip_params) > 0 ? 0 : 1; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback_in_product_package():
    """The product path must not import or call anything under oracle/."""
    pkg = os.path.join(ROOT, "rs-tfhe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_ctx_create_rejects_bad_params_without_gpu():
    from rs_tfhe_amd import _capi

    lib = _capi.lib()
    ctx = ctypes.c_void_p()
    bad = _capi.Params(700, 4, 6, 2, 9)  # l = 4 unsupported
    assert lib.tfhe_hip_ctx_create(ctypes.byref(bad), 0, ctypes.byref(ctx)) == _capi.EINVAL
    assert b"unsupported" in lib.tfhe_hip_last_error(None)
    assert lib.tfhe_hip_ctx_create(None, 0, ctypes.byref(ctx)) == _capi.EINVAL
    lib.tfhe_hip_ctx_destroy(None)  # no-op
    # the concurrent-callers controls: NULL handles are refused, nothing is dereferenced
    st = _capi.CombineStats()
    assert lib.tfhe_hip_set_combining(None, 64) == _capi.EINVAL
    assert lib.tfhe_hip_get_combine_stats(None, ctypes.byref(st)) == _capi.EINVAL


def test_kernel_selectors_are_validated_without_gpu(monkeypatch):
    """TFHE_HIP_BR_KERNEL / TFHE_HIP_KS_KERNEL (the supported controls listed in include/tfhe_hip.h) are checked before
    the library touches a device: unknown values, and kernels the parameter set cannot run, are TFHE_HIP_EINVAL with a
    message; every variable the library reads is in the header's list."""
    from rs_tfhe_amd import _capi

    lib = _capi.lib()
    ctx = ctypes.c_void_p()
    good, uint4 = _capi.Params(700, 3, 6, 2, 9), _capi.Params(820, 1, 22, 5, 3)
    for var, val, params, text in (("TFHE_HIP_KS_KERNEL", "nonsense", good, b"TFHE_HIP_KS_KERNEL must be"),
                                   ("TFHE_HIP_BR_KERNEL", "wide", good, b"TFHE_HIP_BR_KERNEL must be"),
                                   ("TFHE_HIP_KS_KERNEL", "mfma", uint4, b"not available"),
                                   ("TFHE_HIP_KS_KERNEL", "sliced", good, b"not available"),
                                   ("TFHE_HIP_COMBINE", "many", good, b"TFHE_HIP_COMBINE must be"),
                                   ("TFHE_HIP_COMBINE", "5000", good, b"TFHE_HIP_COMBINE must be")):
        monkeypatch.setenv(var, val)
        assert lib.tfhe_hip_ctx_create(ctypes.byref(params), 0, ctypes.byref(ctx)) == _capi.EINVAL, (var, val)
        assert text in lib.tfhe_hip_last_error(None), lib.tfhe_hip_last_error(None)
        monkeypatch.delenv(var)
    # the product library reads exactly the variables the header lists
    src = "".join(open(os.path.join(ROOT, "rs-tfhe_amd", "csrc", f)).read() for f in ("tfhe_hip.hip", "pool.hpp", "combine.hpp"))
    product = re.sub(r"#ifdef TFHE_EXPERIMENT\n(?:(?!#if|#endif).)*#endif\n", "", src, flags=re.S)  # experiment-only overrides (a flat block)
    product = re.sub(r"#ifdef TFHE_EXPERIMENT.*?\n#endif\n#endif\n", "", product, flags=re.S)  # ... (a block with one nested #ifdef)
    read = set(re.findall(r'getenv\("(TFHE_HIP_[A-Z0-9_]+)"\)', product))
    hdr = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    assert read == {"TFHE_HIP_BR_KERNEL", "TFHE_HIP_KS_KERNEL", "TFHE_HIP_POOL_RCCL", "TFHE_HIP_POOL_PINNED_STAGING",
                    "TFHE_HIP_COMBINE"}, read
    for v in read:
        assert v in hdr


def test_pool_create_and_shard_without_gpu():
    """tfhe_hip_pool_*: argument checks and the order-preserving split (rayon_impl.rs:40-47 keeps input order)."""
    from rs_tfhe_amd import _capi
    from rs_tfhe_amd.distributed import shard_range

    lib = _capi.lib()
    pool = ctypes.c_void_p()
    good = _capi.Params(700, 3, 6, 2, 9)
    devs = (ctypes.c_int * 2)(0, 0)
    assert lib.tfhe_hip_pool_create(ctypes.byref(good), devs, 0, ctypes.byref(pool)) == _capi.EINVAL
    assert lib.tfhe_hip_pool_create(ctypes.byref(good), None, 2, ctypes.byref(pool)) == _capi.EINVAL
    bad = _capi.Params(700, 4, 6, 2, 9)
    assert lib.tfhe_hip_pool_create(ctypes.byref(bad), devs, 2, ctypes.byref(pool)) == _capi.EINVAL and not pool.value
    assert lib.tfhe_hip_pool_size(None) == 0 and not lib.tfhe_hip_pool_ctx(None, 0)
    lib.tfhe_hip_pool_destroy(None)
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    for count in (0, 1, 7, 8, 65536, 65537, 524288):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lib.tfhe_hip_pool_shard(count, r, world, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == shard_range(count, r, world) and lo.value == prev
                prev = hi.value
            assert prev == count


def test_param_sets_match_oracle_and_survey(O):
    from rs_tfhe_amd import params as P

    for name, op in O.PARAM_SETS.items():
        pp = P.PARAM_SETS[name]
        assert (pp.n, pp.l, pp.bgbit, pp.basebit, pp.iks_t) == (op.n, op.l, op.bgbit, op.basebit, op.t)
        assert pp.alpha_lv0 == op.alpha_lv0 and pp.alpha_lv1 == op.alpha_lv1
        assert P.gen_decomposition_offset(pp) == O.gen_decomposition_offset(op.l, op.bgbit)
    s = P.SECURITY_128_BIT
    # SURVEY.md section 8 size table
    assert s.tlwe_lv0_bytes == 2804 and s.bsk_bytes == 68_812_800 and s.ksk_bytes == 103_366_656
    assert s.ksk_touched_bytes == 19_381_248
    assert s.algorithmic_bytes_per_bootstrap(2) == 88_202_460
    assert P.SECURITY_UINT4.bsk_bytes == 26_869_760 and P.SECURITY_80_BIT.bsk_bytes == 54_067_200
    for d in (0.125, -0.125, 0.25, -0.25, 1.75, -1.75, 1 / 64, 0.0):
        assert P.f64_to_torus(d) == O.f64_to_torus(d)


def test_param_sets_match_reference_source():
    """Every SecurityParams constant of src/params.rs (11 sets, UINT6 / UINT8 included), parsed from the reference's
    own source text where it lies, equals the product's table (and the C++ mirror's)."""
    path = "/root/reference/src/params.rs"
    if not os.path.exists(path):
        pytest.skip("reference sources not present on this box")
    from rs_tfhe_amd import params as P

    src = re.sub(r"//[^\n]*", "", open(path).read()).replace("_", "")  # comments and digit separators out
    found = {}
    for m in re.finditer(r"pub const (SECURITY[A-Z0-9]+): SecurityParams = SecurityParams \{(.*?)\n\};", src, re.S):
        body = m.group(2)
        lv0 = re.search(r"tlwelv0: TlweParams \{\s*n: (\d+),\s*alpha: ([0-9.e+-]+)", body)
        lv1 = re.search(r"tlwelv1: TlweParams \{\s*n: (\d+),\s*alpha: ([0-9.e+-]+)", body)
        g = re.search(r"trgswlv1: TrgswParams \{(.*?)\}", body, re.S).group(1)
        f = {k: re.search(rf"\b{k}: ([0-9.e+-]+)", g).group(1) for k in ("n", "bgbit", "l", "basebit", "ikst")}
        name = m.group(1).replace("SECURITY", "SECURITY_").replace("BIT", "_BIT")
        found[name] = (int(lv0.group(1)), int(f["l"]), int(f["bgbit"]), int(f["basebit"]), int(f["ikst"]),
                       float(lv0.group(2)), float(lv1.group(2)), int(f["n"]))
    assert len(found) == 11, sorted(found)
    hdr = open(os.path.join(ROOT, "include", "rs_tfhe_hip.hpp")).read()
    for name, (n, l, bgbit, basebit, t, a0, a1, N) in found.items():
        pp = P.PARAM_SETS[name]
        assert (pp.n, pp.l, pp.bgbit, pp.basebit, pp.iks_t, pp.alpha_lv0, pp.alpha_lv1) == (n, l, bgbit, basebit, t, a0, a1), name
        assert N == 1024
        cpp = re.search(rf"constexpr SecurityParams {name}\{{(.*?)\}};", hdr).group(1).split(",")
        assert [int(x) for x in cpp[1:6]] == [n, l, bgbit, basebit, t] and float(cpp[6]) == a0 and float(cpp[7]) == a1, name


def test_lut_generator_matches_oracle(O):
    from rs_tfhe_amd.lut import Encoder, Generator, div_round

    for m, f in ((2, lambda x: x), (2, lambda x: 1 - x), (4, lambda x: (3 * x + 1) % 4), (16, lambda x: (x * x) % 16),
                 (3, lambda x: x), (16, lambda x: x % 16)):
        lut = Generator(m).generate_lookup_table(f)
        assert np.array_equal(lut.poly, O.lut_generate(f, m))
        enc = Encoder(m)
        for x in range(m):
            assert enc.encode(x) == O.lut_encode(x, m)
            assert enc.decode(enc.encode(x)) == x == O.lut_decode(O.lut_encode(x, m), m)
    assert [div_round(10, 3), div_round(11, 3), div_round(12, 3), div_round(1, 2), div_round(0, 5)] == [3, 4, 4, 1, 0]
    g = Generator(4)
    full = g.generate_lookup_table_full(lambda x: Encoder(4).encode(x))
    assert np.array_equal(full.poly, g.generate_lookup_table(lambda x: x).poly)
    assert not Generator(2).generate_lookup_table(lambda x: x).is_empty()  # generator.rs:281-333
    # the _assign forms write into an existing table (generator.rs:89-137, 160-203)
    from rs_tfhe_amd.lut import LookupTable

    t1, t2 = g.generate_lookup_table(lambda x: (3 * x + 1) % 4), LookupTable()
    assert t2.is_empty()
    g.generate_lookup_table_assign(lambda x: (3 * x + 1) % 4, t2)
    assert np.array_equal(t2.poly, t1.poly)
    t2.clear()
    g.generate_lookup_table_full_assign(lambda x: Encoder(4).encode((3 * x + 1) % 4), t2)
    assert np.array_equal(t2.poly, t1.poly)
    assert g.mod_switch(0) == 0 and g.mod_switch(0x80000000) == 512 and g.mod_switch(0xFFFFFFFF) == 0 and g.mod_switch(0x00200000) == 1


def test_shard_range_partitions():
    from rs_tfhe_amd.distributed import shard_counts, shard_range

    for count in (0, 1, 7, 8, 65536, 524288, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(count, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = shard_counts(count, world)
            assert sum(sizes) == count and max(sizes) - min(sizes) <= 1
    assert shard_range(524288, 3, 8) == (196608, 262144)


def test_gates_api_surface():
    """Names and arities of the reference API (src/gates.rs, src/bootstrap/mod.rs)."""
    import inspect

    import rs_tfhe_amd as R

    for name in ("nand", "or_", "and_", "xor", "xnor", "nor", "and_ny", "and_yn", "or_ny", "or_yn"):
        assert len(inspect.signature(getattr(R.Gates, name)).parameters) == 4
        assert callable(getattr(R.gates, name))
    for name in ("mux", "mux_naive"):
        assert len(inspect.signature(getattr(R.Gates, name)).parameters) == 5
    for name in ("batch_nand", "batch_and", "batch_or", "batch_xor", "batch_nor", "batch_xnor", "batch_blind_rotate"):
        assert callable(getattr(R.gates, name))
    for name in ("bootstrap", "bootstrap_without_key_switch", "name"):
        assert name in R.Bootstrap.__abstractmethods__
    assert {"bootstrap_func", "bootstrap_lut"} <= set(dir(R.LutBootstrap))
    g = R.Gates()
    assert g.bootstrap_strategy() == "hip-gfx950"
    a = np.arange(5, dtype=np.uint32)
    assert np.array_equal(g.not_(a), (0 - a.astype(np.int64)).astype(np.uint32))
    assert g.constant(True, 4)[4] == 0x20000000 and g.constant(False, 4)[4] == 0xE0000001  # quirk Q6


def test_kernels_compile_without_scratch_and_keep_their_occupancy():
    """Register-allocation guard: every gfx950 kernel of the library must compile with zero scratch
    (a 12-register spill in the l = 1 blind rotation cost 9 % before it was noticed), the batch blind
    rotation must keep two waves per SIMD and the LDS-ring key switch three."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "rs-tfhe_amd", "csrc", "tfhe_hip.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "--cuda-device-only",
                        "-c", "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", src],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch) == len(occ) and len(names) > 20
    spilled = {n: s for n, s in zip(names, scratch) if s}
    assert not spilled, spilled
    by_name = dict(zip(names, occ))
    for n, o in by_name.items():
        if "14k_blind_rotateI" in n:
            assert o >= 2, (n, o)
        if "k_key_switch_b4" in n:
            assert o >= 3, (n, o)


def test_keyed_engine_pool_logic(monkeypatch):
    """bootstrap.keyed_engine (host logic, no GPU): ONE context per (parameter set, device); every cloud key gets a
    key view of it that stays resident (no re-upload), at most MAX_RESIDENT_KEYS views are kept, the least recently
    used IDLE one is dropped beyond that, and a view in use is never dropped."""
    import threading

    from rs_tfhe_amd import bootstrap as B
    from rs_tfhe_amd.params import SECURITY_128_BIT as P

    contexts, views = [], []

    class FakeLib:
        @staticmethod
        def tfhe_hip_key_is_loaded(ctx):
            return int(ctx.loaded)

    class FakeEngine:
        def __init__(self, params, device, _view_of=None):
            self.params, self.device, self._parent = params, device, _view_of
            self._key, self._last_use, self._users, self.lock = None, 0, 0, threading.RLock()
            self._lib, self._ctx = FakeLib, self
            self.loaded, self.loads, self.closed = False, 0, False
            (views if _view_of is not None else contexts).append(self)

        def new_key_view(self):
            return FakeEngine(self.params, self.device, _view_of=self)

        def load_cloud_key(self, ck):
            self._key, self.loaded = ck, True
            self.loads += 1

        def close(self):
            self.closed = True

    class Key:
        params = P

    monkeypatch.setattr(B, "Engine", FakeEngine)
    monkeypatch.setattr(B, "_engines", {})
    monkeypatch.setattr(B, "_views", {})
    keys = [Key() for _ in range(6)]
    with B.keyed_engine(keys[0]) as v0:
        assert v0._key is keys[0] and v0._parent is contexts[0] and v0._users == 1
    assert v0._users == 0
    with B.keyed_engine(keys[0]) as again:
        assert again is v0 and v0.loads == 1  # no re-upload
    held = []
    for k in keys[1:4]:
        with B.keyed_engine(k) as v:
            held.append(v)
    assert len(contexts) == 1 and len(views) == B.MAX_RESIDENT_KEYS == 4  # one context, four resident keys
    with B.keyed_engine(keys[0]):  # touch key 0: key 1's view is now the least recently used
        pass
    with B.keyed_engine(keys[4]) as v:
        assert v._key is keys[4] and held[0].closed and not v0.closed and len(contexts) == 1
    with B.keyed_engine(keys[0]) as v:
        assert v is v0 and v0.loads == 1  # still resident
    # a view in use is never the victim, whatever its age
    with B.keyed_engine(keys[2]) as busy:
        for k in (keys[1], keys[5], Key()):
            with B.keyed_engine(k):
                pass
        assert not busy.closed
    # another device gets its own context
    with B.keyed_engine(keys[5], device=1) as v:
        assert v.device == 1 and len(contexts) == 2
    assert B.engine_for(P, 0) is contexts[0]


def test_adopted_views_count_against_the_resident_key_cap(monkeypatch):
    """bootstrap.adopt_view (what SecretKey.cloud_key() registers a freshly generated key with) shares the LRU eviction
    of keyed_engine: a loop of key generations keeps at most MAX_RESIDENT_KEYS views resident (each pins 276 MB of device
    memory and its CloudKey on the host) instead of growing without bound; a view in use is never dropped, and
    re-adopting the same key object replaces its idle stale view."""
    import threading

    from rs_tfhe_amd import bootstrap as B
    from rs_tfhe_amd.params import SECURITY_128_BIT as P

    class FakeView:
        def __init__(self):
            self.params, self.device = P, 0
            self._key, self._last_use, self._users, self.lock = None, 0, 0, threading.RLock()
            self.closed = False

        def close(self):
            self.closed = True

    monkeypatch.setattr(B, "_engines", {})
    monkeypatch.setattr(B, "_views", {})

    class Key:
        params = P

    keys, views = [Key() for _ in range(7)], [FakeView() for _ in range(7)]
    for k, v in zip(keys[:4], views[:4]):
        B.adopt_view(k, v)
    assert len(B._views[(P, 0)]) == 4 and not any(v.closed for v in views[:4])
    views[1]._users = 1  # in use: must survive
    B.adopt_view(keys[4], views[4])  # evicts the least recently used idle view: views[0]
    assert views[0].closed and not views[1].closed and len(B._views[(P, 0)]) == 4
    B.adopt_view(keys[5], views[5])  # views[1] is in use: views[2] goes
    assert views[2].closed and not views[1].closed
    for i in range(40):  # key rotation: bounded
        B.adopt_view(Key(), FakeView())
        assert len(B._views[(P, 0)]) <= B.MAX_RESIDENT_KEYS
    assert not views[1].closed
    stale, fresh = FakeView(), FakeView()
    B.adopt_view(keys[6], stale)
    B.adopt_view(keys[6], fresh)  # the same CloudKey object generated into another view
    assert stale.closed and B._views[(P, 0)][id(keys[6])] is fresh


def test_profiles_readme_counter_block_is_generated_from_the_entries():
    """profiles/README.md's per-launch counter paragraphs are the output of profiles/readme_counters.py over
    profiles/pmc_roofline.json (round 3's README printed round 2's HBM figure for the `r3` entry): no drift."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("readme_counters", os.path.join(ROOT, "profiles", "readme_counters.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    assert mod.block() in text, "run `python3 profiles/readme_counters.py --install`"


def test_measured_tables_are_generated_from_the_committed_bench_lines():
    """DESIGN.md section 5 and profiles/README.md carry the output of profiles/readme_bench.py over the committed
    profiles/r4_*bench*.json[l] files, and the headline figures quoted in README.md / DESIGN.md's summary are those of
    profiles/r4_bench.json: the documents cannot drift from the evidence they cite."""
    import importlib.util
    import re

    spec = importlib.util.spec_from_file_location("readme_bench", os.path.join(ROOT, "profiles", "readme_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    block = mod.block()
    for doc in ("DESIGN.md", os.path.join("profiles", "README.md")):
        assert block in open(os.path.join(ROOT, doc)).read(), f"{doc}: run `python3 profiles/readme_bench.py --install`"
    hp = mod.headline_paragraph()
    for doc in ("README.md", "DESIGN.md"):
        assert hp in open(os.path.join(ROOT, doc)).read(), f"{doc}: run `python3 profiles/readme_bench.py --install`"


def test_bench_picks_a_free_rendezvous_port():
    """bench.py's torchrun child gets a port that is free at launch (the driver's 1 -> 8 sweep starts bench.py four times
    in a row on one node; a fixed port can meet the previous run's lingering listener)."""
    import socket

    import bench

    held = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    held.bind(("127.0.0.1", 0))
    held.listen(1)
    try:
        ports = {bench.free_port() for _ in range(8)}
        assert held.getsockname()[1] not in ports and all(1024 < p < 65536 for p in ports)
        for p in ports:  # and each can be bound right away
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                s.bind(("127.0.0.1", p))
    finally:
        held.close()
