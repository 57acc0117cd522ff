#!/bin/bash
# After `gpurun -- bash profiles/exp/evidence_r6.sh`: install what came back under gpurun_out/ into profiles/ and
# regenerate the generated blocks of the documents (the headline figures in README.md / DESIGN.md prose are edited by
# hand; tests/test_host_logic.py says which).
cd "$(dirname "$0")/.."
python3 profiles/install_entries.py r6 r6_uint4 r6_mixed80
for f in r6_bench.json r6_uint4_bench.json r6_80bit_xor_bench.json r6_mixed80_bench.json r6_other_sets_bench.jsonl \
         r6_pool_resident_2ctx.json r6_pool_resident_8ctx.json r6_pool_resident_mixed80.json r6_pool_resident_rccl_loopback.json \
         r6_pool_host_1ctx.json r6_pool_host_pinned.json r6_stage_bench.jsonl r6_concurrent_calls.jsonl r6_concurrent_calls_uint4.jsonl; do
  cp gpurun_out/$f profiles/$f
done
python3 profiles/readme_counters.py --install
python3 profiles/readme_bench.py --install
python3 -c "
import importlib.util, os
s = importlib.util.spec_from_file_location('rb', 'profiles/readme_bench.py'); m = importlib.util.module_from_spec(s); s.loader.exec_module(m)
print('headline: %.1f k, blind rotate %.1f ms, key switch %.2f ms, frac (algorithmic) %.3f, frac_executed %.3f' % ((m.headline()[0] / 1e3,) + m.headline()[1:]))"
