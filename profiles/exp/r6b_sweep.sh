#!/bin/bash
# round 6: the combining front end under a team of host threads -- repeat runs, the pool, lane counts, linger settings
O=gpurun_out/r6b; mkdir -p $O
for i in 1 2 3; do python3 profiles/exp/concurrent_calls.py --threads 1,8,16,64,256,512,1024 --seconds 0.4 2>&1 | cut -c1-170; done > $O/conc3.log 2>&1
python3 profiles/exp/concurrent_calls.py --pool 0,0 --threads 8,64,256,1024 --seconds 0.4 2>&1 | cut -c1-170 > $O/pool.log
export TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so
for lanes in 1 3; do echo lanes $lanes; TFHE_HIP_COMBINE_LANES=$lanes python3 profiles/exp/concurrent_calls.py --threads 8,64,256,512 --seconds 0.3 2>&1 | cut -c1-170; done > $O/lanes.log 2>&1
for q in 10 50 100; do echo quiet $q; TFHE_HIP_LINGER_QUIET_US=$q python3 profiles/exp/concurrent_calls.py --threads 8,64,256,512 --seconds 0.3 2>&1 | cut -c1-170; done > $O/quiet.log 2>&1
unset TFHE_HIP_LIB TFHE_HIP_ALLOW_EXPERIMENT
timeout 900 python -m pytest tests/test_gpu_combine.py -x -q -s > $O/test_combine.log 2>&1
tail -8 $O/test_combine.log
