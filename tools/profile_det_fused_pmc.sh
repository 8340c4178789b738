# Counters for the detector's fused kernels (csrc/det_fused.h): four PMC passes over a few forwards of 16 pages, each in its own run with
# --kernel-trace only (gpurun refuses --pmc beside the other trace domains).
#   gpurun --timeout 900 -- 'bash tools/profile_det_fused_pmc.sh [det_fuse value, default 31]'
# Output: gpurun_out/detf_pmc_set<k>.md (tools/rocpd_pmc.py --raw: mean raw value per kernel and counter instance).
FUSE=${1:-31}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/det_op_times.py --fuse $FUSE --reps 1 --steps 1"
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format rocpd -d /tmp/dfp$i -- $CMD > /tmp/dfp$i.out 2> /tmp/dfp$i.err
  i=$((i + 1))
done
cd $R
db() { find /tmp/$1 -name "*.db" | head -1; }
for k in 1 2 3 4; do
  d=$(db dfp$k)
  if [ -n "$d" ]; then python tools/rocpd_pmc.py --raw $d > gpurun_out/detf_pmc_set$k.md 2>&1; else tail -5 /tmp/dfp$k.err > gpurun_out/detf_pmc_set$k.md; fi
done
grep -c "" gpurun_out/detf_pmc_set*.md
