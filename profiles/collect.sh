#!/bin/bash
# Collect the rocprofv3 evidence for one tag on the GPU box:  bash profiles/collect.sh <tag> [bench args]
# One --kernel-trace pass, then separate --pmc passes (never combined with traces), reduced to
# gpurun_out/<tag>_kernel_stats.csv / <tag>_pmc.csv by summarize_rocpd.py and to a gpurun_out/<tag>_pmc_roofline.json
# entry (per-launch HBM bytes and issue fractions of the two kernels) by pmc_roofline.py; copy those into
# profiles/ (the entry goes into profiles/pmc_roofline.json, which bench.py reads for `roofline.traffic`).
# COLLECT_PROG="profiles/exp/latency.py --counts 1" bash profiles/collect.sh r3_lat   profiles another program of the
# repo (a python script; no bench args are added and no pmc_roofline entry is made).
tag=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
if [ -n "$COLLECT_PROG" ]; then TRACE_CMD="$COLLECT_PROG"; PMC_CMD="$COLLECT_PROG"
else TRACE_CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $*"; PMC_CMD="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs $*"; fi
rocprofv3 --kernel-trace --stats -d "$O/trace" -- python3 $TRACE_CMD > "$O/trace.log" 2>&1
# the bench line of THIS run (HIP-event launch times of the very launches the trace holds) beside the trace summary
grep '^{"metric"' "$O/trace.log" > "$R/gpurun_out/${tag}_trace_bench.json" || true
i=0
while read -r counters; do
  i=$((i+1))
  rocprofv3 --pmc $counters -d "$O/pmc$i" -- python3 $PMC_CMD > "$O/pmc$i.log" 2>&1
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
LIST
PROFILE_OUT="$R/gpurun_out" python3 profiles/summarize_rocpd.py "$tag" $(find "$O/trace" -name '*.db' | head -1) $(find "$O"/pmc* -name '*.db') > "$O/summary.txt" 2>&1
[ -n "$COLLECT_PROG" ] || python3 profiles/pmc_roofline.py "$R/gpurun_out/${tag}_pmc.csv" "$R/gpurun_out/${tag}_kernel_stats.csv" "$tag" "$@" > "$R/gpurun_out/${tag}_pmc_roofline.json" 2>> "$O/summary.txt"
tail -5 "$O/summary.txt"
