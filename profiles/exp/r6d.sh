#!/bin/bash
# round 6: whole GPU suite on the one-lane front end; the mid-size two-stream experiment; the multi-device tests as a dry run on {0, 0}
O=gpurun_out/r6d; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -5 $O/gpu_suite.log
TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so python3 profiles/exp/midsize.py > $O/midsize.log 2>&1
TFHE_HIP_TEST_DEVICES=0,0 timeout 1500 python -m pytest tests/test_gpu_multi_device.py -x -q -s -m gpu > $O/multi_dry.log 2>&1
tail -5 $O/multi_dry.log
