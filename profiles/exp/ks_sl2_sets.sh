#!/bin/bash
# k_key_switch_sliced2 at every instantiated number of accumulator sets (experiment build: TFHE_HIP_KS_SLICED_SETS), 65,536
# ciphertexts; the digest must be the same in every line.  usage: ks_sl2_sets.sh [PARAMS...]
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R"
for P in ${@:-SECURITY_UINT4}; do
  for S in ${SETS:-auto 24 28 32 36}; do
    echo -n "$P sets=$S: "
    if [ $S = auto ]; then unset TFHE_HIP_KS_SLICED_SETS; else export TFHE_HIP_KS_SLICED_SETS=$S; fi
    TFHE_HIP_LIB=$R/rs-tfhe_amd/${LIB:-libtfhe_v_sl2.so} python3 profiles/exp/ks_only.py --params $P --reps 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['key_switch_ms'], d['max_board_w'], d['digest'])"
  done
done
