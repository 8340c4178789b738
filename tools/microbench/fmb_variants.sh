#!/bin/bash
# A/B builds of fmb_kernel (csrc/det_fused.h: SA_FMB_PK packed chunk epilogue, SA_FMB_NB3 three-buffer ring, SA_FMB_OVL next tile's patch under the
# tail) into tools/microbench/abl/ (git-ignored *.so, travels with gpurun).   usage: tools/microbench/fmb_variants.sh 000 100 010 001 ...  (PK NB3 OVL)
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  SURYA_AMD_CXXFLAGS="-DSA_FMB_PK=${v:0:1} -DSA_FMB_NB3=${v:1:1} -DSA_FMB_OVL=${v:2:1}" SURYA_AMD_LIB_OUT="$PWD/tools/microbench/abl/libfmbv_$v.so" python -m surya_amd.build --force > /tmp/fmbv_$v.log 2>&1 &
  if (( $(jobs -r | wc -l) >= 3 )); then wait -n; fi
done
wait
ls -la tools/microbench/abl/
