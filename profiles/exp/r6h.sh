#!/bin/bash
# lane prepared at key load; zero-copy inputs A/B (experiment build); whole GPU suite
O=gpurun_out/r6h; mkdir -p $O
export TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so
for z in 0 1 0 1; do echo zero_copy $z; TFHE_HIP_COMBINE_ZEROCOPY=$z python3 profiles/exp/phases.py 2>&1 | grep -v amdgpu; done > $O/zerocopy.log 2>&1
unset TFHE_HIP_LIB TFHE_HIP_ALLOW_EXPERIMENT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
