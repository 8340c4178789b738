#!/bin/bash
# Ablation builds of fmb_kernel (csrc/det_fused.h, SA_FMB_ABL bits) into tools/microbench/abl/ (git-ignored *.so, travels with gpurun).
# usage: tools/microbench/dwproj_ablate.sh 1 2 4 8 ...   then on the GPU box: for v in ...; SURYA_AMD_LIB=tools/microbench/abl/libfmb_$v.so python tools/det_op_times.py --fuse 16
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  SURYA_AMD_CXXFLAGS="-DSA_FMB_ABL=$v" SURYA_AMD_LIB_OUT="$PWD/tools/microbench/abl/libfmb_$v.so" python -m surya_amd.build --force > /tmp/abl_$v.log 2>&1 &
  if (( $(jobs -r | wc -l) >= 2 )); then wait -n; fi
done
wait
ls -la tools/microbench/abl/
