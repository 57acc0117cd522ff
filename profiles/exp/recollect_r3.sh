bash profiles/collect.sh r3 > gpurun_out/r3_collect.log 2>&1
bash profiles/collect.sh r3_uint4 --params SECURITY_UINT4 --gate pbs > gpurun_out/r3_uint4_collect.log 2>&1
bash profiles/collect.sh r3_mixed80 --params SECURITY_80_BIT --gate mixed --batch 131072 > gpurun_out/r3_mixed80_collect.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
python3 bench.py --steps 10 --warmup 3 --params SECURITY_UINT4 --gate pbs --no-cpu-baseline > gpurun_out/r3_uint4_bench.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --params SECURITY_80_BIT --gate xor --no-cpu-baseline > gpurun_out/r3_80bit_xor_bench.json 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --params SECURITY_80_BIT --gate mixed --batch 131072 --no-cpu-baseline > gpurun_out/r3_mixed80_bench.json 2>/dev/null
tail -2 gpurun_out/r3_collect.log; head -c 600 gpurun_out/r3_bench.json
