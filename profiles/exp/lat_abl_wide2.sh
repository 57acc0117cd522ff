for v in ${VARIANTS:-l8 l16 l32 l64 l128 lall}; do TFHE_HIP_ALLOW_EXPERIMENT=1 TFHE_HIP_LIB=rs-tfhe_amd/libtfhe_v_$v.so python3 profiles/exp/latency_ablation.py $v 2>&1 | grep -v amdgpu.ids | head -1; done
python3 profiles/exp/latency_ablation.py base | head -1
