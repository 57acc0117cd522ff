#!/bin/bash
# res.sh "<flags>" : VGPRs / scratch / occupancy of the blind-rotation kernels for a knob setting (compile only)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R/rs-tfhe_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function $1 \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null tfhe_hip.hip 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - \
  | grep -E "k_blind_rotateILi[123]ELb1|k_external_productILi3ELb1" | sed -E 's/Function Name: _ZN4tfhe//'
