# Counters for the detector's short-K convolution GEMMs ("after round 4"): they run at ~0.55 PF/s AND ~2.1 TB/s at
# once, and tile size / the persistent loop / 128 x 128 tiles change nothing -- so, counters before variants. Four PMC passes over two
# forwards of 16 pages (tools/microbench/det_sweep.py's model, one configuration), each in its own run with --kernel-trace only.
#   gpurun --timeout 600 -- 'bash tools/profile_det_pmc.sh'
# Output: gpurun_out/det_pmc_set<k>.md (tools/rocpd_pmc.py --raw: mean raw value per kernel); copy what is cited into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --det-only --no-cpu-baseline --det-steps 2"
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format rocpd -d /tmp/dp$i -- $CMD > /tmp/dp$i.out 2> /tmp/dp$i.err
  i=$((i + 1))
done
cd $R
db() { find /tmp/$1 -name "*.db" | head -1; }
for k in 1 2 3 4; do
  d=$(db dp$k)
  if [ -n "$d" ]; then python tools/rocpd_pmc.py --raw $d > gpurun_out/det_pmc_set$k.md 2>&1; else tail -5 /tmp/dp$k.err > gpurun_out/det_pmc_set$k.md; fi
done
grep -c "" gpurun_out/det_pmc_set*.md
