# Step 1 of the round's evidence refresh (on the GPU box): traces + counters for the three profiled workloads.
# Step 2 (here): python3 profiles/install_entries.py r6 r6_uint4 r6_mixed80.  Step 3 (GPU box): profiles/exp/rebench_r6.sh.
bash profiles/collect.sh r6 > gpurun_out/r6_collect.log 2>&1
bash profiles/collect.sh r6_uint4 --params SECURITY_UINT4 --gate pbs > gpurun_out/r6_uint4_collect.log 2>&1
bash profiles/collect.sh r6_mixed80 --params SECURITY_80_BIT --gate mixed --batch 131072 > gpurun_out/r6_mixed80_collect.log 2>&1
tail -2 gpurun_out/r6_collect.log
