#!/bin/bash
# round 6: front end with ONE lane and the want-scaled linger; two lanes again under a kernel trace (which hardware
# queues the lanes' launches ran on, in the order that showed the bad mode: one thread first, then 64); whole GPU suite
O=gpurun_out/r6c; mkdir -p $O
for i in 1 2 3; do python3 profiles/exp/concurrent_calls.py --threads 1,8,16,64,256,512,1024 --seconds 0.4 2>&1 | cut -c1-170; done > $O/conc3.log 2>&1
python3 profiles/exp/concurrent_calls.py --pool 0,0 --threads 8,64,256,1024 --seconds 0.4 2>&1 | cut -c1-170 > $O/pool.log
export TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=$GRAFT_REPO_ROOT/rs-tfhe_amd/libtfhe_v_comb.so
for q in 0 75; do echo quiet $q; TFHE_HIP_LINGER_QUIET_US=$q python3 profiles/exp/concurrent_calls.py --threads 8,64,256,512 --seconds 0.3 2>&1 | cut -c1-170; done > $O/quiet.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TFHE_HIP_COMBINE_LANES=2 rocprofv3 --kernel-trace -d $O/trace -- python3 profiles/exp/concurrent_calls.py --threads 1,64 --seconds 0.3 > $O/trace.log 2>&1
python3 profiles/exp/overlap.py $(find $O/trace -name "*.db" | head -1) > $O/overlap_1_then_64.txt 2>&1; rm -rf $O/trace
grep '"threads"' $O/trace.log | cut -c1-170 >> $O/overlap_1_then_64.txt
unset TFHE_HIP_LIB TFHE_HIP_ALLOW_EXPERIMENT
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -5 $O/gpu_suite.log
