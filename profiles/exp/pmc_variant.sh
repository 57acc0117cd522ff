#!/bin/bash
# Cache-path counters of one kernel variant:  bash profiles/exp/pmc_variant.sh <tag> <lib.so> [counter lists file]
# (separate --pmc passes of profiles/exp/ab.py --child under TFHE_HIP_LIB; summaries -> gpurun_out/<tag>_pmc.csv)
tag=${1:?tag}; lib=${2:?lib}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
export TFHE_HIP_LIB=$R/$lib
i=0
while read -r counters; do
  i=$((i+1))
  rocprofv3 --pmc $counters -d "$O/pmc$i" -- python3 profiles/exp/ab.py --child --steps 1 > "$O/pmc$i.log" 2>&1
done <<'LIST'
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
LIST
python3 - "$tag" "$O" <<'PY'
import glob, sqlite3, sys
tag, O = sys.argv[1], sys.argv[2]
out = open(f"{O}/../{tag}_pmc.csv", "w")
out.write("kernel,counter,dispatches,sum,per_launch\n")
for db in sorted(glob.glob(f"{O}/pmc*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for k, c, n, v in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name"):
        if "blind_rotate" in k or "key_switch" in k:
            out.write(f"{k[:60]},{c},{n},{v:.1f},{v/max(n,1):.1f}\n")
PY
cat "$O/../${tag}_pmc.csv"
