# key switch alone by batch size: split kernel (matrix cores off), matrix-core kernel without / with K chunks
for B in 32 64 128 256 512 1024 2048 4096; do
  for cfg in "TFHE_HIP_KS_KERNEL=split" "TFHE_HIP_KS_KERNEL=mfma TFHE_HIP_KS_MFMA_KSPLIT=1" "TFHE_HIP_KS_KERNEL=mfma"; do  # (KSPLIT: experiment builds, TFHE_HIP_LIB=libtfhe_v_base.so)
    env $cfg python3 profiles/exp/ks_only.py --batch $B --reps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$B', '$cfg'.ljust(48), d['key_switch_ms'], d['digest'])"
  done
done
