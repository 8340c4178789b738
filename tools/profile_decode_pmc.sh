# Next measurement for the decode chain (lead 2): every structural hypothesis about the 18 us gate|up launch
# was measured and refuted (profiles/r02_decode_sweeps.md), so the next step is counters, not variants. One kernel-trace pass + four
# PMC passes over 16 decode steps of the shipped configuration (tools/microbench/decode_sweep.py --configs base), each PMC pass in
# its own run with --kernel-trace only (gpurun refuses --pmc together with the other trace domains). ~25 s of GPU per pass.
#   gpurun --timeout 400 -- 'bash tools/profile_decode_pmc.sh'
# Output: gpurun_out/decode_pmc_<set>.md (tools/rocpd_pmc.py tables per kernel); copy what is cited into profiles/.
# Counter names: the ones MI355X_MICROARCH.md / round 1 (profiles/r01_pmc_sq_counters.md) showed to exist on gfx950; a pass whose
# names rocprofv3 rejects fails by itself and leaves the others intact.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/microbench/decode_sweep.py --configs base --steps 16"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/dq0 -- $CMD > /tmp/dq0.out 2> /tmp/dq0.err
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  rocprofv3 --kernel-trace --pmc $set --output-format rocpd -d /tmp/dq$i -- $CMD > /tmp/dq$i.out 2> /tmp/dq$i.err
  i=$((i + 1))
done
cd $R
db() { find /tmp/$1 -name "*.db" | head -1; }
python tools/rocpd_stats.py $(db dq0) --by-grid > gpurun_out/decode_pmc_kernel_stats.md 2>&1
for k in 1 2 3 4; do
  d=$(db dq$k)
  if [ -n "$d" ]; then python tools/rocpd_pmc.py --raw $d > gpurun_out/decode_pmc_set$k.md 2>&1; else tail -5 /tmp/dq$k.err > gpurun_out/decode_pmc_set$k.md; fi
done
head -12 gpurun_out/decode_pmc_kernel_stats.md | cut -c1-160
