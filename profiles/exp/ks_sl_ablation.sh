#!/bin/bash
# Where the column-sliced key switch (SECURITY_UINT4, 65,536 ciphertexts) spends its time: timing-only builds with
# parts removed (TFHE_ABL_SL bits, csrc/experiment.hpp; results wrong by construction).  Run on the GPU box.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R"
export TFHE_HIP_ALLOW_EXPERIMENT=1
for v in base nodma nobar norestage nolds nodma_nobar all; do
  lib=rs-tfhe_amd/libtfhe_v_sl_$v.so
  [ "$v" = base ] && lib=rs-tfhe_amd/libtfhe_hip.so
  echo -n "$v: "
  TFHE_HIP_LIB=$R/$lib python3 profiles/exp/ks_only.py --params SECURITY_UINT4 --reps 4 | cut -c1-400
done
