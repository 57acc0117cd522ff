# key switch alone at SECURITY_UINT4 by batch size: split kernel / sliced kernel whole walk / sliced kernel with K chunks (shipped)
for B in 32 64 128 256 512 1024 2048 4096 8192; do
  for cfg in "TFHE_HIP_KS_SL_CHUNK_MIN=100000000" "TFHE_HIP_KS_SL_KCHUNKS=1 TFHE_HIP_KS_SPLIT_MAX=0" "TFHE_HIP_KS_SL_CHUNK_MIN=1"; do
    env $cfg python3 profiles/exp/ks_only.py --params SECURITY_UINT4 --batch $B --reps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$B', '$cfg'.ljust(52), d['key_switch_ms'], d['digest'])"
  done
done
