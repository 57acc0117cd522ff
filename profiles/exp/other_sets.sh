# bench lines of the parameter sets BASELINE does not quote (gates on the boolean sets, pbs at the set's own modulus)
: > gpurun_out/r3_other_sets_bench.jsonl
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_110_BIT --gate nand >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_UINT1 --gate pbs --modulus 2 >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_UINT2 --gate pbs --modulus 4 >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_UINT3 --gate pbs --modulus 8 >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_UINT5 --gate pbs --modulus 16 >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --params SECURITY_128_BIT --gate mux_naive >> gpurun_out/r3_other_sets_bench.jsonl 2>/dev/null
wc -l gpurun_out/r3_other_sets_bench.jsonl
