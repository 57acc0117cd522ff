// C++ host-mirror tests, written after the reference's own unit tests
// (src/gates.rs:559-681, 832-858; src/bootstrap/vanilla.rs:78-141; src/bootstrap/lut.rs:142-254).
// Key material and decryption come from the CPU oracle (test infrastructure); everything
// under test goes through include/rs_tfhe_hip.hpp -> the C ABI -> the HIP kernels.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>

#include "rs_tfhe_hip.hpp"

extern "C" {
typedef struct {
  int32_t n, l, bgbit, basebit, t;
  double alpha_lv0, alpha_lv1;
} orc_params;
void orc_init(void);
void orc_gen_secret_key(uint64_t seed, int n, uint32_t *key_lv0, uint32_t *key_lv1);
void orc_gen_bootstrapping_key(uint64_t seed, const orc_params *P, const uint32_t *k0, const uint32_t *k1,
                               double *bsk_fft, uint32_t *bsk_time);
void orc_gen_key_switching_key(uint64_t seed, const orc_params *P, const uint32_t *k0, const uint32_t *k1,
                               uint32_t *ksk);
void orc_tlwe_encrypt_f64(uint64_t seed, double p, double alpha, const uint32_t *key, int dim, uint32_t *out);
int orc_tlwe_decrypt_bool(const uint32_t *ct, const uint32_t *key, int dim);
int orc_tlwe_decrypt_lwe_message(const uint32_t *ct, int m, const uint32_t *key, int dim);
typedef struct {
  orc_params P;
  uint32_t decomposition_offset;
  const uint32_t *testvec;
  const double *bsk_fft;
  const uint32_t *bsk_time;
  const uint32_t *ksk;
} orc_cloud_key;
int orc_batch_gate(const orc_cloud_key *ck, int op, const uint32_t *a, const uint32_t *b, uint32_t *out, int count, int nthreads);
void orc_reencrypt_tlwe_lv0(const uint32_t *src, const uint32_t *key, const orc_params *P, uint32_t *out);
// the four HIP runtime calls the device-resident pool test needs (libamdhip64 is linked; no HIP headers under g++)
int hipMalloc(void **p, size_t bytes);
int hipFree(void *p);
int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);  // 1 = host to device, 2 = device to host
int hipDeviceSynchronize(void);
}

using namespace rs_tfhe;

static int failures = 0;
#include <chrono>
static std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static void lap(const char *what) {  // where the test's time goes (printed, not asserted)
  const auto t = std::chrono::steady_clock::now();
  std::printf("[%6.1f s] %s\n", std::chrono::duration<double>(t - g_t0).count(), what);
  g_t0 = t;
}
#define CHECK(cond, ...)                              \
  do {                                                \
    if (!(cond)) {                                    \
      ++failures;                                     \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                       \
      std::printf("\n");                              \
    }                                                 \
  } while (0)

struct OracleSecretKey {  // key::SecretKey (src/key.rs:21-49), filled by the oracle's generator
  std::vector<Torus> key_lv0, key_lv1;
};

static uint64_t g_seed = 1;
static Ciphertext encrypt_bool(bool b, const SecurityParams &P, const OracleSecretKey &sk) {  // tlwe.rs:55-58
  Ciphertext c(P.n);
  orc_tlwe_encrypt_f64(g_seed++, b ? 0.125 : -0.125, P.alpha_lv0, sk.key_lv0.data(), P.n, c.p.data());
  return c;
}
static Ciphertext encrypt_lwe_message(size_t m, size_t modulus, const SecurityParams &P, const OracleSecretKey &sk) {  // tlwe.rs:84-98
  Ciphertext c(P.n);
  orc_tlwe_encrypt_f64(g_seed++, (double)(m % modulus) / (2.0 * modulus), P.alpha_lv0, sk.key_lv0.data(), P.n, c.p.data());
  return c;
}
static bool decrypt_bool(const Ciphertext &c, const OracleSecretKey &sk) {
  return orc_tlwe_decrypt_bool(c.p.data(), sk.key_lv0.data(), c.n()) != 0;
}

// gates.rs:832-858 `test_gate`
static void test_gate(const char *name, const std::function<bool(bool, bool)> &expect,
                      const std::function<Ciphertext(const Gates &, const Ciphertext &, const Ciphertext &, const CloudKey &)> &actual,
                      const OracleSecretKey &key, const CloudKey &cloud_key) {
  Gates gates;
  const bool cases[4][2] = {{true, true}, {true, false}, {false, true}, {false, false}};
  for (auto &tc : cases) {
    Ciphertext ct_a = encrypt_bool(tc[0], cloud_key.params, key);
    Ciphertext ct_b = encrypt_bool(tc[1], cloud_key.params, key);
    Ciphertext result = actual(gates, ct_a, ct_b, cloud_key);
    CHECK(decrypt_bool(result, key) == expect(tc[0], tc[1]), "%s failed for %d %d", name, tc[0], tc[1]);
  }
}

int main() {
  orc_init();
  const SecurityParams P = SECURITY_128_BIT;
  orc_params OP{P.n, P.l, P.bgbit, P.basebit, P.iks_t, P.alpha_lv0, P.alpha_lv1};
  OracleSecretKey key;
  key.key_lv0.resize(P.n);
  key.key_lv1.resize(N);
  orc_gen_secret_key(99, P.n, key.key_lv0.data(), key.key_lv1.data());
  CloudKey cloud_key;
  cloud_key.params = P;
  cloud_key.decomposition_offset = gen_decomposition_offset(P);
  cloud_key.blind_rotate_testvec = gen_testvec();
  cloud_key.bootstrapping_key.resize((size_t)P.n * 2 * P.l * 2 * N);
  cloud_key.key_switching_key.resize(N * (size_t)P.iks_t * P.base() * (P.n + 1));
  orc_gen_bootstrapping_key(199, &OP, key.key_lv0.data(), key.key_lv1.data(), cloud_key.bootstrapping_key.data(), nullptr);
  orc_gen_key_switching_key(200, &OP, key.key_lv0.data(), key.key_lv1.data(), cloud_key.key_switching_key.data());
  CHECK(cloud_key.decomposition_offset == 0x82080000u, "decomposition offset");

  // gates.rs:559-653
  test_gate("nand", [](bool a, bool b) { return !(a & b); }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.nand(a, b, k); }, key, cloud_key);
  test_gate("or", [](bool a, bool b) { return a | b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.or_(a, b, k); }, key, cloud_key);
  test_gate("and", [](bool a, bool b) { return a & b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.and_(a, b, k); }, key, cloud_key);
  test_gate("xor", [](bool a, bool b) { return a ^ b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.xor_(a, b, k); }, key, cloud_key);
  test_gate("xnor", [](bool a, bool b) { return false ^ (b ^ a); },  // sic, gates.rs:576-582
            [](const Gates &g, auto &a, auto &b, auto &k) { return g.xnor(a, b, k); }, key, cloud_key);
  test_gate("nor", [](bool a, bool b) { return !(a | b); }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.nor(a, b, k); }, key, cloud_key);
  test_gate("and_ny", [](bool a, bool b) { return !a & b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.and_ny(a, b, k); }, key, cloud_key);
  test_gate("and_yn", [](bool a, bool b) { return a & !b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.and_yn(a, b, k); }, key, cloud_key);
  test_gate("or_ny", [](bool a, bool b) { return !a | b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.or_ny(a, b, k); }, key, cloud_key);
  test_gate("or_yn", [](bool a, bool b) { return a | !b; }, [](const Gates &g, auto &a, auto &b, auto &k) { return g.or_yn(a, b, k); }, key, cloud_key);
  test_gate("not", [](bool a, bool) { return !a; }, [](const Gates &g, auto &a, auto &, auto &) { return g.not_(a); }, key, cloud_key);
  test_gate("copy", [](bool a, bool) { return a; }, [](const Gates &g, auto &a, auto &, auto &) { return g.copy(a); }, key, cloud_key);

  // gates.rs:656-681 test_mux (mux_naive)
  {
    Gates gates;
    for (int v = 0; v < 8; ++v) {
      bool a = v & 4, b = v & 2, c = v & 1;
      Ciphertext r = gates.mux_naive(encrypt_bool(a, P, key), encrypt_bool(b, P, key), encrypt_bool(c, P, key), cloud_key);
      CHECK(decrypt_bool(r, key) == (a ? b : c), "mux_naive %d %d %d", a, b, c);
    }
  }
  // gates.rs:352-383 batch_nand
  {
    gates::Pairs in;
    const bool cases[4][2] = {{true, true}, {true, false}, {false, true}, {false, false}};
    for (int rep = 0; rep < 3; ++rep)
      for (auto &tc : cases) in.push_back({encrypt_bool(tc[0], P, key), encrypt_bool(tc[1], P, key)});
    auto out = gates::batch_nand(in, cloud_key);
    CHECK(out.size() == in.size(), "batch size");
    for (size_t i = 0; i < out.size(); ++i) CHECK(decrypt_bool(out[i], key) == !(cases[i % 4][0] && cases[i % 4][1]), "batch_nand %zu", i);
    CHECK(gates::batch_nand({}, cloud_key).empty(), "empty batch");
  }
  // vanilla.rs:78-98 / :127-141 bootstrap through the trait object
  {
    std::unique_ptr<Bootstrap> bs = default_bootstrap();
    CHECK(bs->name() == "hip-gfx950", "strategy name %s", bs->name().c_str());
    for (bool v : {true, false}) CHECK(decrypt_bool(bs->bootstrap(encrypt_bool(v, P, key), cloud_key), key) == v, "bootstrap %d", v);
    Ciphertext h = bs->bootstrap_without_key_switch(encrypt_bool(true, P, key), cloud_key);  // vanilla.rs:100-125: no-panic only
    CHECK(h.n() == P.n, "bootstrap_without_key_switch shape");
  }
  // lut.rs:142-254 identity / NOT / constant / LUT reuse, message_modulus = 2
  {
    LutBootstrap lb;
    for (size_t m : {0u, 1u}) {
      Ciphertext ct = encrypt_lwe_message(m, 2, P, key);
      auto dec = [&](const Ciphertext &c) { return (size_t)orc_tlwe_decrypt_lwe_message(c.p.data(), 2, key.key_lv0.data(), P.n); };
      CHECK(dec(lb.bootstrap_func(ct, [](size_t x) { return x; }, 2, cloud_key)) == m, "lut identity");
      CHECK(dec(lb.bootstrap_func(ct, [](size_t x) { return 1 - x; }, 2, cloud_key)) == 1 - m, "lut not");
      CHECK(dec(lb.bootstrap_func(ct, [](size_t) { return (size_t)1; }, 2, cloud_key)) == 1, "lut const");
      lut::LookupTable table = lut::Generator(2).generate_lookup_table([](size_t x) { return 1 - x; });
      CHECK(dec(lb.bootstrap_lut(ct, table, cloud_key)) == 1 - m, "lut reuse");
    }
    CHECK(lb.name() == "lut-hip-gfx950", "lut name");
  }
  // error behaviour: the reference panics; here a std::runtime_error
  {
    bool threw = false;
    try {
      Gates().nand(Ciphertext(3), Ciphertext(3), cloud_key);
    } catch (const std::runtime_error &) {
      threw = true;
    }
    CHECK(threw, "dimension mismatch must throw");
  }
  // trgsw.rs:289-294
  {
    auto r = trgsw::batch_blind_rotate({encrypt_bool(true, P, key), encrypt_bool(false, P, key)}, cloud_key);
    CHECK(r.size() == 2, "batch_blind_rotate size");
  }
  // client side through the header alone (key.rs:21-66, tlwe.rs:37-126): keys, GPU key generation,
  // encryption, a gate, decryption -- cross-checked against the oracle's decoders on the same words
  {
    rs_tfhe::SecretKey sk = rs_tfhe::SecretKey::generate(P, 4242);
    CHECK(sk.key_lv0.size() == (size_t)P.n && sk.key_lv1.size() == N, "secret key sizes");
    CloudKey gk = generate_cloud_key_seeded(sk, 77);
    CHECK(gk.decomposition_offset == 0x82080000u, "generated key: decomposition offset");
    CHECK(gk.blind_rotate_testvec.b[5] == 0x20000000u && gk.blind_rotate_testvec.a[5] == 0, "generated key: test vector");
    ChaChaRng rng(5);
    Gates g;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        Ciphertext ca = tlwe::encrypt_bool(a, P.alpha_lv0, sk.key_lv0, rng), cb = tlwe::encrypt_bool(b, P.alpha_lv0, sk.key_lv0, rng);
        CHECK(tlwe::decrypt_bool(ca, sk.key_lv0) == (bool)a, "fresh encryption decrypts");
        CHECK(tlwe::decrypt_bool(ca, sk.key_lv0) == (orc_tlwe_decrypt_bool(ca.p.data(), sk.key_lv0.data(), P.n) != 0), "decoders agree");
        Ciphertext r = g.nand(ca, cb, gk);
        CHECK(tlwe::decrypt_bool(r, sk.key_lv0) == !(a && b), "nand under a GPU-generated key %d %d", a, b);
        CHECK(tlwe::decrypt_bool(r, sk.key_lv0) == (orc_tlwe_decrypt_bool(r.p.data(), sk.key_lv0.data(), P.n) != 0), "decoders agree on the result");
      }
    for (size_t m = 0; m < 4; ++m) {
      Ciphertext c = tlwe::encrypt_lwe_message(m + 8, 4, P.alpha_lv0, sk.key_lv0, rng);
      CHECK(tlwe::decrypt_lwe_message(c, 4, sk.key_lv0) == m, "message round trip %zu", m);
      CHECK((int)tlwe::decrypt_lwe_message(c, 4, sk.key_lv0) == orc_tlwe_decrypt_lwe_message(c.p.data(), 4, sk.key_lv0.data(), P.n), "message decoders agree");
    }
    // TLWE arithmetic (tlwe.rs:129-214) and the first step of examples/lut_add_two_numbers.rs:
    // bootstrap_lut(&(&x + &y), &lut): 1 + 2 = 3 under modulus 4
    {
      Ciphertext x = tlwe::encrypt_lwe_message(1, 4, P.alpha_lv0, sk.key_lv0, rng), y = tlwe::encrypt_lwe_message(2, 4, P.alpha_lv0, sk.key_lv0, rng);
      CHECK(tlwe::decrypt_lwe_message(x + y, 4, sk.key_lv0) == 3, "x + y");
      CHECK(tlwe::decrypt_lwe_message((x + y) - y, 4, sk.key_lv0) == 1, "(x + y) - y");
      CHECK(tlwe::decrypt_lwe_message(add_mul(x, x, 2), 4, sk.key_lv0) == 3, "add_mul");
      Ciphertext z = -x;
      CHECK(z.p[0] == 0u - x.p[0] && z.b() == 0u - x.b(), "neg");
      lut::Generator gen(4);
      LutBootstrap lb;
      Ciphertext r = lb.bootstrap_lut(x + y, gen.generate_lookup_table([](size_t v) { return (v + 1) % 4; }), gk);
      CHECK(tlwe::decrypt_lwe_message(r, 4, sk.key_lv0) == 0, "bootstrap_lut(x + y)");
      // the rest of the generator / lookup-table surface (generator.rs:89-137, 160-222, 235-238; lookup_table.rs:33-68)
      lut::LookupTable t1 = gen.generate_lookup_table([](size_t v) { return (3 * v + 1) % 4; }), t2;
      CHECK(t2.is_empty() && !t1.is_empty(), "LookupTable::is_empty");
      gen.generate_lookup_table_assign([](size_t v) { return (3 * v + 1) % 4; }, t2);
      CHECK(t2.poly.b == t1.poly.b && t2.poly.a == t1.poly.a, "generate_lookup_table_assign");
      lut::LookupTable t3 = gen.generate_lookup_table_full([&](size_t v) { return lut::Encoder(4).encode((3 * v + 1) % 4); });
      CHECK(t3.poly.b == t1.poly.b, "generate_lookup_table_full with the encoder's values = generate_lookup_table");
      gen.generate_lookup_table_full_assign([&](size_t v) { return lut::Encoder(4).encode((3 * v + 1) % 4); }, t2);
      CHECK(t2.poly.b == t1.poly.b, "generate_lookup_table_full_assign");
      CHECK(lut::Generator(2).generate_lookup_table_custom([](size_t v) { return (3 * v + 1) % 4; }, 4, 1.0 / 8.0).poly.b == t1.poly.b, "generate_lookup_table_custom");
      CHECK(lut::LookupTable::from_poly(t1.poly).poly.b == t1.poly.b, "LookupTable::from_poly");
      t2.clear();
      CHECK(t2.is_empty(), "LookupTable::clear");
      CHECK(gen.mod_switch(0) == 0 && gen.mod_switch(0x80000000u) == 512 && gen.mod_switch(0xFFFFFFFFu) == 0 && gen.mod_switch(0x00200000u) == 1,
            "mod_switch (generator.rs:235-238)");
    }
    // proxy re-encryption through the header alone (src/proxy_reenc.rs): Alice -> Bob symmetric, and through Bob's
    // public key; the GPU's words equal reencrypt_tlwe_lv0 (:468-510) of the oracle on the same key
    {
      rs_tfhe::SecretKey bob = rs_tfhe::SecretKey::generate(P, 4343);
      ChaChaRng prng(6);
      proxy_reenc::ProxyReencryptionKey rk = proxy_reenc::ProxyReencryptionKey::new_symmetric(sk.key_lv0, bob, prng);
      CHECK(rk.key_encryptions.size() == (size_t)P.n * P.iks_t * P.base() * (P.n + 1), "re-encryption key size (proxy_reenc.rs:645-649)");
      std::vector<Ciphertext> cts;
      std::vector<bool> msg;
      for (int i = 0; i < 70; ++i) {
        msg.push_back((prng() & 1u) != 0);
        cts.push_back(tlwe::encrypt_bool(msg.back(), P.alpha_lv0, sk.key_lv0, prng));
      }
      std::vector<Ciphertext> got = rk.reencrypt(cts);
      int ok = 0, same = 0;
      for (size_t i = 0; i < cts.size(); ++i) {
        ok += tlwe::decrypt_bool(got[i], bob.key_lv0) == msg[i];
        Ciphertext want(P.n);
        orc_reencrypt_tlwe_lv0(cts[i].p.data(), rk.key_encryptions.data(), &OP, want.p.data());
        same += want.p == got[i].p;
      }
      CHECK(ok == (int)cts.size(), "symmetric re-encryption decrypts under the target key: %d of %zu", ok, cts.size());
      CHECK(same == (int)cts.size(), "re-encryption bit-identical to the oracle: %d of %zu", same, cts.size());
      CHECK(proxy_reenc::reencrypt_tlwe_lv0(cts[3], rk).p == got[3].p, "single-ciphertext form");
      proxy_reenc::PublicKeyLv0 pub = proxy_reenc::PublicKeyLv0::generate(bob, prng);
      CHECK(pub.encryptions.size() == (size_t)2 * P.n, "public key: 2n encryptions of zero");
      int pk_ok = 0;
      for (int i = 0; i < 40; ++i) pk_ok += tlwe::decrypt_bool(pub.encrypt_bool(i & 1, P.alpha_lv0, prng), bob.key_lv0) == (bool)(i & 1);
      CHECK(pk_ok > 36, "public-key encryption decrypts (> 90 %%, the reference's bar): %d of 40", pk_ok);
      proxy_reenc::ProxyReencryptionKey ak = proxy_reenc::ProxyReencryptionKey::new_asymmetric(sk.key_lv0, pub, prng);
      std::vector<Ciphertext> agot = ak.reencrypt(cts);
      int aok = 0;
      for (size_t i = 0; i < cts.size(); ++i) aok += tlwe::decrypt_bool(agot[i], bob.key_lv0) == msg[i];
      CHECK(aok * 10 > (int)cts.size() * 9, "asymmetric re-encryption (> 90 %%, proxy_reenc.rs:627-632): %d of %zu", aok, cts.size());
      lap("proxy re-encryption");
    }
    // a different key object at the same address must not be mistaken for the loaded one
    CloudKey *slot = new CloudKey(gk);
    CHECK(tlwe::decrypt_bool(g.nand(tlwe::encrypt_bool(true, P.alpha_lv0, sk.key_lv0, rng), tlwe::encrypt_bool(true, P.alpha_lv0, sk.key_lv0, rng), *slot), sk.key_lv0) == false, "copy of the key");
    *slot = cloud_key;  // the oracle-generated key for `key`, same address
    CHECK(decrypt_bool(g.nand(encrypt_bool(true, P, key), encrypt_bool(false, P, key), *slot), key) == true, "address reuse re-uploads");
    delete slot;

    lap("gates, bootstraps, LUTs, GPU keygen, TLWE arithmetic");
    // ---- two threads, two keys, alternating (bootstrap/mod.rs:23: strategies are Send + Sync and every call
    // names its &CloudKey): each thread must always compute under ITS key.  A shared context whose key is
    // checked and then used in two steps fails this; one key view per key on ONE context passes it.
    {
      rs_tfhe::SecretKey sk2 = rs_tfhe::SecretKey::generate(P, 999);
      CloudKey gk2 = generate_cloud_key_seeded(sk2, 78);
      std::atomic<int> bad{0};
      auto worker = [&](const rs_tfhe::SecretKey &s, const CloudKey &k, uint64_t seed) {
        ChaChaRng r(seed);
        Gates gg;
        for (int it = 0; it < 6; ++it) {
          const bool a = (it & 1) != 0, b = (it & 2) != 0;
          Ciphertext ca = tlwe::encrypt_bool(a, P.alpha_lv0, s.key_lv0, r), cb = tlwe::encrypt_bool(b, P.alpha_lv0, s.key_lv0, r);
          if (tlwe::decrypt_bool(gg.nand(ca, cb, k), s.key_lv0) != !(a && b)) ++bad;
          gates::Pairs in{{ca, cb}, {cb, ca}, {ca, ca}};
          auto out = gates::batch_xor(in, k);
          if (tlwe::decrypt_bool(out[0], s.key_lv0) != (a != b) || tlwe::decrypt_bool(out[2], s.key_lv0) != false) ++bad;
        }
      };
      std::thread t1(worker, std::cref(sk), std::cref(gk), 11), t2(worker, std::cref(sk2), std::cref(gk2), 12);
      t1.join();
      t2.join();
      CHECK(bad.load() == 0, "two threads with two alternating keys: %d wrong results", bad.load());
      // ... and they did it on ONE context: both keys are key views (tfhe_hip_key_create) of the same handle
      Engine &base = Engine::for_params(P, 0);
      auto b1 = Engine::for_key(gk), b2 = Engine::for_key(gk2);
      CHECK(b1.context() == base.ctx() && b2.context() == base.ctx() && b1.handle() != b2.handle(), "one context, two key views");
      CHECK(tfhe_hip_key_parent(b1.handle()) == base.ctx() && tfhe_hip_key_is_loaded(b2.handle()) == 1, "views of the shared context");
      CHECK(base.resident_keys() >= 2 && base.resident_keys() <= Engine::kMaxResidentKeys, "resident key views: %zu", base.resident_keys());
    }
    lap("two threads, two keys");
    // several devices behind one handle (here: two contexts on GPU 0): same words as the single-context path,
    // in input order, at a count that does not divide evenly
    {
      DevicePool pool(P, {0, 0});
      CHECK(pool.size() == 2, "pool size");
      pool.load(gk);
      ChaChaRng r(21);
      gates::Pairs in;
      std::vector<bool> want;
      for (int i = 0; i < 7; ++i) {
        const bool a = (i % 3) == 0, b = (i & 1) != 0;
        in.push_back({tlwe::encrypt_bool(a, P.alpha_lv0, sk.key_lv0, r), tlwe::encrypt_bool(b, P.alpha_lv0, sk.key_lv0, r)});
        want.push_back(!(a && b));
      }
      auto one = gates::batch_nand(in, gk);
      auto two = pool.batch_gate(TFHE_HIP_NAND, in);
      bool same = one.size() == two.size();
      for (size_t i = 0; same && i < one.size(); ++i) same = one[i].p == two[i].p && tlwe::decrypt_bool(two[i], sk.key_lv0) == want[i];
      CHECK(same, "pool of two contexts equals the single-context batch word for word");
    }
    lap("pool of two contexts, host pointers");
    // ... and for a batch that is RESIDENT on one member's GPU (tfhe_hip_pool_batch_*_dev): device pointers in, the
    // shards travel between the members, device pointer out, input order kept; home = the second member
    {
      DevicePool pool(P, {0, 0});
      pool.load(gk);
      ChaChaRng r(22);
      const size_t count = 601, w = (size_t)P.n + 1;
      gates::Pairs in;
      for (size_t i = 0; i < count; ++i)
        in.push_back({tlwe::encrypt_bool((i % 3) == 0, P.alpha_lv0, sk.key_lv0, r), tlwe::encrypt_bool((i & 1) != 0, P.alpha_lv0, sk.key_lv0, r)});
      std::vector<Torus> fa(count * w), fb(count * w), got(count * w);
      for (size_t i = 0; i < count; ++i) {
        std::copy(in[i].first.p.begin(), in[i].first.p.end(), fa.begin() + i * w);
        std::copy(in[i].second.p.begin(), in[i].second.p.end(), fb.begin() + i * w);
      }
      void *da = nullptr, *db = nullptr, *dout = nullptr;
      const bool alloc = hipMalloc(&da, count * w * 4) == 0 && hipMalloc(&db, count * w * 4) == 0 && hipMalloc(&dout, count * w * 4) == 0;
      CHECK(alloc, "hipMalloc");
      if (alloc) {
        hipMemcpy(da, fa.data(), count * w * 4, 1);
        hipMemcpy(db, fb.data(), count * w * 4, 1);
        pool.batch_gate_dev(1, TFHE_HIP_NAND, (const Torus *)da, (const Torus *)db, (Torus *)dout, count);
        pool.synchronize();
        hipDeviceSynchronize();
        hipMemcpy(got.data(), dout, count * w * 4, 2);
        auto host = pool.batch_gate(TFHE_HIP_NAND, in);
        bool same = true;
        for (size_t i = 0; same && i < count; ++i) same = std::equal(host[i].p.begin(), host[i].p.end(), got.begin() + i * w);
        CHECK(same, "device-resident pool call equals the host-pointer pool call word for word");
        CHECK(std::string(pool.data_transport()) == "peer-copy", "transport of a pool that repeats a device: %s", pool.data_transport());
      }
      hipFree(da);
      hipFree(db);
      hipFree(dout);
    }
    lap("pool of two contexts, device-resident batch");
    // tfhe_hip_last_error is per (thread, handle): `Bootstrap: Send + Sync` (bootstrap/mod.rs:23) lets two threads
    // share one context; the thread that fails must read ITS text while the other keeps working (and sees none)
    {
      auto view = Engine::for_key(gk);
      tfhe_hip_ctx *h = view.handle();
      ChaChaRng r(23);
      Ciphertext ca = tlwe::encrypt_bool(true, P.alpha_lv0, sk.key_lv0, r);
      std::atomic<int> bad{0};
      std::thread failing([&] {
        std::vector<Torus> out(P.n + 1);
        for (int it = 0; it < 60; ++it) {
          const int rc = tfhe_hip_batch_gate(h, 99, ca.p.data(), ca.p.data(), out.data(), 1);
          if (rc != TFHE_HIP_EINVAL || std::string(tfhe_hip_last_error(h)) != "unknown gate") ++bad;
          const int rc2 = tfhe_hip_batch_sample_extract(h, nullptr, 5000, nullptr, 1);
          if (rc2 != TFHE_HIP_EINVAL || std::string(tfhe_hip_last_error(h)) != "extraction index out of range") ++bad;
        }
      });
      std::thread working([&] {  // a bounded number of gates: std::mutex is not fair, an endless worker could starve the other thread
        std::vector<Torus> out(P.n + 1);
        for (int it = 0; it < 40; ++it) {
          if (tfhe_hip_batch_gate(h, TFHE_HIP_NAND, ca.p.data(), ca.p.data(), out.data(), 1) != TFHE_HIP_OK) ++bad;
          if (std::string(tfhe_hip_last_error(h)) != "") ++bad;  // another thread's failure is not this thread's
        }
      });
      failing.join();
      working.join();
      CHECK(bad.load() == 0, "per-thread error text: %d mismatches", bad.load());
    }
    lap("per-thread error text");
    // ---- a team of threads calling ONE strategy (`Bootstrap: Send + Sync`, bootstrap/mod.rs:23; what
    // `pairs.par_iter().map(|(a, b)| gates.nand(a, b, ck))` does with Rayon's workers): the library merges the calls
    // that are in flight together into shared launches (combine.hpp).  Every result must be the CPU path's word for
    // word, and the team must get many times what one thread gets.
    {
      const int T = 32, K = 12;
      std::vector<Ciphertext> A, B, R((size_t)T * K, Ciphertext(P.n));
      std::vector<int> op((size_t)T * K);
      for (int i = 0; i < T * K; ++i) {
        A.push_back(encrypt_bool((i * 7 + 1) % 3 == 0, P, key));
        B.push_back(encrypt_bool((i * 5 + 2) % 4 < 2, P, key));
        op[(size_t)i] = i % 3;  // nand, xor, and
      }
      Gates warm;
      (void)warm.nand(A[0], B[0], cloud_key);  // the oracle-generated key is resident before the clock starts
      auto one = [&](const Gates &g, int i) {
        return op[(size_t)i] == 0 ? g.nand(A[(size_t)i], B[(size_t)i], cloud_key)
               : op[(size_t)i] == 1 ? g.xor_(A[(size_t)i], B[(size_t)i], cloud_key)
                                    : g.and_(A[(size_t)i], B[(size_t)i], cloud_key);
      };
      auto t0 = std::chrono::steady_clock::now();
      {
        Gates g;
        for (int i = 0; i < 24; ++i) R[(size_t)i] = one(g, i);
      }
      const double alone = 24 / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> team;
      for (int t = 0; t < T; ++t)
        team.emplace_back([&, t] {
          Gates g;
          for (int i = t * K; i < (t + 1) * K; ++i) R[(size_t)i] = one(g, i);
        });
      for (auto &th : team) th.join();
      const double together = T * K / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("one thread %.0f gates/s, a team of %d threads %.0f gates/s\n", alone, T, together);
      CHECK(together > 8 * alone, "a team of %d threads gets %.0f gates/s, one thread %.0f", T, together, alone);
      orc_cloud_key ock{OP, cloud_key.decomposition_offset, cloud_key.blind_rotate_testvec.a.data(), cloud_key.bootstrapping_key.data(),
                        nullptr, cloud_key.key_switching_key.data()};
      const int codes[3] = {TFHE_HIP_NAND, TFHE_HIP_XOR, TFHE_HIP_AND};
      int wrong = 0;
      for (int c = 0; c < 3; ++c) {
        std::vector<Torus> fa, fb;
        std::vector<int> idx;
        for (int i = c; i < T * K; i += 3) {
          idx.push_back(i);
          fa.insert(fa.end(), A[(size_t)i].p.begin(), A[(size_t)i].p.end());
          fb.insert(fb.end(), B[(size_t)i].p.begin(), B[(size_t)i].p.end());
        }
        std::vector<Torus> fo(fa.size());
        CHECK(orc_batch_gate(&ock, codes[c], fa.data(), fb.data(), fo.data(), (int)idx.size(), 0) == 0, "oracle gate");
        for (size_t q = 0; q < idx.size(); ++q)
          if (!std::equal(R[(size_t)idx[q]].p.begin(), R[(size_t)idx[q]].p.end(), fo.begin() + (long)(q * (size_t)(P.n + 1)))) ++wrong;
      }
      CHECK(wrong == 0, "team of threads: %d of %d results differ from the CPU path", wrong, T * K);
    }
    lap("a team of threads on one strategy");
    // OS-keyed generation (the default): a usable key, different every time
    {
      rs_tfhe::SecretKey sk3 = rs_tfhe::SecretKey::generate(P);
      CloudKey k3 = generate_cloud_key(sk3), k4 = generate_cloud_key(sk3);
      CHECK(k3.key_switching_key != k4.key_switching_key, "OS-keyed generation must differ from call to call");
      ChaChaRng r3;
      Ciphertext ca = tlwe::encrypt_bool(true, P.alpha_lv0, sk3.key_lv0, r3), cb = tlwe::encrypt_bool(true, P.alpha_lv0, sk3.key_lv0, r3);
      CHECK(tlwe::decrypt_bool(Gates().nand(ca, cb, k3), sk3.key_lv0) == false, "nand under an OS-keyed cloud key");
      ChaChaRng a1(7), a2(7), a3(8);
      CHECK(a1() == a2() && a1() != a3(), "ChaChaRng(seed) is reproducible");
    }
  }
  lap("OS-keyed generation");
  std::printf(failures ? "%d FAILURES\n" : "all C++ mirror tests passed (%d failures)\n", failures);
  return failures ? 1 : 0;
}
