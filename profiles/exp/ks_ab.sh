#!/bin/bash
# Key switch alone, several builds of the library interleaved: ks_ab.sh PARAMS ROUNDS lib1.so lib2.so ...
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R"
P=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$R/rs-tfhe_amd/$lib python3 profiles/exp/ks_only.py --params $P --reps 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['key_switch_ms'], d['max_board_w'], d['digest'])"
  done
done
