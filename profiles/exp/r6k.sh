#!/bin/bash
O=gpurun_out/r6k; mkdir -p $O
export TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so
for q in 25 50 100 25 50 100; do echo quiet $q; TFHE_HIP_LINGER_QUIET_US=$q python3 profiles/exp/concurrent_calls.py --threads 4,8,16,32,64,256 --seconds 0.5 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    s = d['stats']
    print(d['threads'], d['gates_per_s'], 'median', d['call_ms_median'], 'p99', d['call_ms_p99'], 'per launch', round(s['requests'] / max(1, s['launches']), 2))
"; done > $O/quiet.log 2>&1
