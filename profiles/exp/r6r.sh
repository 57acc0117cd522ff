#!/bin/bash
# bulk launches cut into chunks (TFHE_HIP_BR_CHUNK, experiment build): what a small call waits, and what the bulk path pays
O=gpurun_out/r6r; mkdir -p $O
export TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so
for c in 0 16384 8192 4096 2048; do
  echo "chunk $c"
  TFHE_HIP_BR_CHUNK=$c python3 profiles/exp/mixed_load.py 2>&1 | grep '"bulk_batch": 65536' 
  TFHE_HIP_BR_CHUNK=$c python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done > $O/chunks.log 2>&1
cat $O/chunks.log
