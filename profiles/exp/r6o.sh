#!/bin/bash
# final refresh of round 6's evidence on the final tree + the wall time of the driver's own command
O=gpurun_out; mkdir -p $O
bash profiles/exp/evidence_r6.sh > $O/r6_evidence.log 2>&1
tail -2 $O/r6_evidence.log | cut -c1-300
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_driver_cmd.json 2> $O/r6_driver_cmd.time
grep -E "Elapsed|Maximum resident" $O/r6_driver_cmd.time
