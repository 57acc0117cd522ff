#!/bin/bash
# The round's whole evidence refresh in ONE gpurun call, so that the traces, the counters and the bench lines come from
# the same box (boxes differ by up to 6 % under the board's power cap):
#   recollect_r6.sh (traces + counters) -> install_entries.py (on the box's copy) -> rebench_r6.sh (bench lines)
# Afterwards, here: python3 profiles/install_entries.py r6 r6_uint4 r6_mixed80; python3 profiles/readme_counters.py --install;
# copy gpurun_out/r6_*bench*.json* into profiles/.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash profiles/exp/recollect_r6.sh
python3 profiles/install_entries.py r6 r6_uint4 r6_mixed80
bash profiles/exp/rebench_r6.sh
