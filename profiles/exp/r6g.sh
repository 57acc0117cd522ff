#!/bin/bash
O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
python3 profiles/exp/phases.py > $O/phases.log 2>&1
