#!/bin/bash
# Ablation builds of mbconv_kernel (csrc/det_mbconv.h, SA_MBC_ABL bits) into tools/microbench/abl/ (git-ignored *.so, travels with gpurun).
# usage: tools/microbench/mbconv_ablate.sh 1 2 4 8 ...   then on the GPU box: for v in ...; SURYA_AMD_LIB=tools/microbench/abl/libmbc_$v.so python tools/det_op_times.py --fuse 127
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  SURYA_AMD_CXXFLAGS="-DSA_MBC_ABL=$v" SURYA_AMD_LIB_OUT="$PWD/tools/microbench/abl/libmbc_$v.so" python -m surya_amd.build --force > /tmp/abl_$v.log 2>&1 &
  if (( $(jobs -r | wc -l) >= 2 )); then wait -n; fi
done
wait
ls -la tools/microbench/abl/
