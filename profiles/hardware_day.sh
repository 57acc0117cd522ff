#!/bin/bash
# Hardware day: everything that needs SEVERAL MI355X (or a Rust toolchain), in one go.  Run from the repository root on
# a box with N >= 2 GPUs:      bash profiles/hardware_day.sh [/path/to/rs-tfhe checkout]
# Writes profiles/hw_day/*.json(l) and a pass / fail summary; needs no edits (the device list is every GPU of the box).
#   1. the contract bench at N = 1, 2, 4, 8 (one process per GPU over RCCL, key broadcast, no data-path collective);
#      the N > 1 lines carry BASELINE configs[2] through ONE pool handle (`pool_resident`)
#   2. that pool-resident leg on its own over distinct devices: shards by grouped ncclSend / ncclRecv, then by peer copies
#      (key replication both ways is in the same records: key_transport, key_replication_s, comm_create_s)
#   3. tests/test_gpu_multi_device.py (asserts the expected ranges: >= 7.5 x at 8 GPUs, scatter <= 10 ms and gather <= 5 ms
#      per peer for 367.5 / 183.8 MB) and tests/test_gpu_pool_resident.py
#   4. with a Rust toolchain and a checkout of thedonutfactory/rs-tfhe: the binding compiled and its tests run
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/profiles/hw_day; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 __graft_entry__.py || exit 1
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "GPUs: $NGPU"
DEVS=$(python3 -c "print(','.join(str(i) for i in range($NGPU)))")
: > "$O/bench_scaling.jsonl"
for n in 1 2 4 8; do
  [ "$n" -le "$NGPU" ] || continue
  python3 bench.py --gpus $n --steps 20 --warmup 5 $([ "$n" -gt 1 ] && echo --no-cpu-baseline) | grep '^{' | tee -a "$O/bench_scaling.jsonl" | cut -c1-200
done
if [ "$NGPU" -ge 2 ]; then
  TFHE_HIP_POOL_RCCL=1 python3 bench.py --pool-devices $DEVS --resident --steps 5 --warmup 2 --oracle-sample 16 | grep '^{' > "$O/pool_resident_rccl.json"
  TFHE_HIP_POOL_RCCL=0 python3 bench.py --pool-devices $DEVS --resident --steps 5 --warmup 2 --oracle-sample 16 | grep '^{' > "$O/pool_resident_peer_copy.json"
  TFHE_HIP_POOL_RCCL=1 python3 bench.py --pool-devices $DEVS --resident --steps 5 --warmup 2 --gate mixed --params SECURITY_80_BIT --batch 131072 | grep '^{' > "$O/pool_resident_configs4_rccl.json"
  python3 bench.py --pool-devices $DEVS --steps 3 --warmup 1 | grep '^{' > "$O/pool_host_pageable.json"
  python3 bench.py --pool-devices $DEVS --steps 3 --warmup 1 --pinned | grep '^{' > "$O/pool_host_pinned.json"
fi
python3 -m pytest tests/test_gpu_multi_device.py tests/test_gpu_pool_resident.py -q -m gpu -s 2>&1 | tee "$O/pytest.log" | tail -15
python3 - "$O" "$NGPU" <<'PY'
import json, sys, os
O, n = sys.argv[1], int(sys.argv[2])
rows = [json.loads(l) for l in open(os.path.join(O, "bench_scaling.jsonl")) if l.startswith("{")]
by = {r["n_gpus"]: r for r in rows}
ok = True
for k in sorted(by):
    eff = by[k]["value"] / (k * by[1]["value"]) if 1 in by else float("nan")
    print(f"N = {k}: {by[k]['value']:.0f} bootstraps/s, {by[k]['ms_per_step']} ms per step, efficiency {eff:.3f}"
          + (f", pool_resident {by[k]['pool_resident']}" if by[k].get("pool_resident") else ""))
    ok &= k == 1 or eff >= 0.9375
for name in ("pool_resident_rccl", "pool_resident_peer_copy"):
    p = os.path.join(O, name + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        d = json.load(open(p))
        print(name, {k: d[k] for k in ("transport", "key_transport", "value", "scatter_ms", "gather_ms", "scatter_group_ms", "gather_group_ms", "comm_create_s", "key_replication_s", "decrypt_ok", "oracle_sample_equal")})
        ok &= d["decrypt_ok"] and d["oracle_sample_equal"] and d["scatter_ms"] <= 10.0 and d["gather_ms"] <= 5.0
print("HARDWARE DAY:", "PASS" if ok and n >= 2 else ("needs >= 2 GPUs" if n < 2 else "FAIL"))
PY
if [ $# -ge 1 ] && command -v cargo > /dev/null; then
  sh rust/apply.sh "$1" && ( cd "$1" && TFHE_HIP_LIB_DIR=$R/rs-tfhe_amd LD_LIBRARY_PATH=$R/rs-tfhe_amd cargo test --release --features "hip lut-bootstrap proxy-reenc" 2>&1 | tee "$O/cargo_test.log" | tail -20 )
else
  echo "rust: skipped (give the path of an rs-tfhe checkout; needs cargo)  --  sh rust/apply.sh <crate> && cargo test --release --features \"hip lut-bootstrap proxy-reenc\""
fi
