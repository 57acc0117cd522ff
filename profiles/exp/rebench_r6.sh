# bench lines of round 6 (GPU box; after install_entries.py so that `roofline.traffic` quotes this round's counters)
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
python3 bench.py --steps 10 --warmup 3 --params SECURITY_UINT4 --gate pbs --no-cpu-baseline > gpurun_out/r6_uint4_bench.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --params SECURITY_80_BIT --gate xor --no-cpu-baseline > gpurun_out/r6_80bit_xor_bench.json 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --params SECURITY_80_BIT --gate mixed --batch 131072 --no-cpu-baseline > gpurun_out/r6_mixed80_bench.json 2>/dev/null
# the parameter sets BASELINE does not quote (gates on the boolean sets, pbs at the set's own modulus; UINT5..8 at 16: N stays 1024)
: > gpurun_out/r6_other_sets_bench.jsonl
for a in "SECURITY_110_BIT --gate nand" "SECURITY_UINT1 --gate pbs --modulus 2" "SECURITY_UINT2 --gate pbs --modulus 4" \
         "SECURITY_UINT3 --gate pbs --modulus 8" "SECURITY_UINT5 --gate pbs --modulus 16" "SECURITY_UINT6 --gate pbs --modulus 16" \
         "SECURITY_UINT7 --gate pbs --modulus 16" "SECURITY_UINT8 --gate pbs --modulus 16" "SECURITY_128_BIT --gate mux_naive"; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params $a >> gpurun_out/r6_other_sets_bench.jsonl 2>/dev/null
done
# several members behind one handle, batch resident on the first: peer-copy transport on this one-GPU box
python3 bench.py --pool-devices 0,0 --resident --steps 3 --warmup 1 --batch 32768 2>/dev/null | grep '^{' > gpurun_out/r6_pool_resident_2ctx.json
python3 bench.py --pool-devices 0,0,0,0,0,0,0,0 --resident --steps 3 --warmup 1 --batch 8192 2>/dev/null | grep '^{' > gpurun_out/r6_pool_resident_8ctx.json
python3 bench.py --pool-devices 0,0 --resident --gate mixed --params SECURITY_80_BIT --steps 3 --warmup 1 --batch 65536 2>/dev/null | grep '^{' > gpurun_out/r6_pool_resident_mixed80.json
TFHE_HIP_POOL_RCCL=2 python3 bench.py --pool-devices 0 --resident --steps 3 --warmup 1 --batch 65536 2>/dev/null | grep '^{' > gpurun_out/r6_pool_resident_rccl_loopback.json
python3 bench.py --pool-devices 0 --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r6_pool_host_1ctx.json
python3 bench.py --pool-devices 0 --pinned --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r6_pool_host_pinned.json
# the reference's criterion groups `bootstrapping` and `fft_operations`
: > gpurun_out/r6_stage_bench.jsonl
for st in blind_rotate ifft fft poly_mul; do python3 bench.py --stage $st --steps 5 --warmup 2 2>/dev/null | grep '^{' >> gpurun_out/r6_stage_bench.jsonl; done
head -c 300 gpurun_out/r6_bench.json
# the combining front end under a team of host threads (one context; a pool of two members on this GPU)
python3 profiles/exp/concurrent_calls.py --threads 1,2,4,8,16,32,64,128,256,512,1024 --seconds 0.5 --off 2>/dev/null | grep '^{' > gpurun_out/r6_concurrent_calls.jsonl
python3 profiles/exp/concurrent_calls.py --params SECURITY_UINT4 --threads 1,8,64,256 --seconds 0.4 2>/dev/null | grep '^{' > gpurun_out/r6_concurrent_calls_uint4.jsonl
