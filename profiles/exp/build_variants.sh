#!/bin/bash
# Build kernel variants of libtfhe_hip.so for profiles/exp/ab.py:  build_variants.sh name "-DKNOB=1 ..." [name flags]...
# Output: rs-tfhe_amd/libtfhe_v_<name>.so (git-ignored, travels to the GPU box with the snapshot).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R/rs-tfhe_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -DTFHE_EXPERIMENT $flags \
      -Rpass-analysis=kernel-resource-usage -shared -o ../libtfhe_v_$name.so tfhe_hip.hip 2>&1 \
      | grep -A8 "k_blind_rotateILi3ELb1E" | grep -E "VGPRs:|ScratchSize|SGPRs:" | tr '\n' ' ' | sed "s/^/$name: /"; echo ) &
done
wait
