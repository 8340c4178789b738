# VERDICT r05 item 3, first half: the counter pass on the big-tile GEMM shapes of the recognition path BEFORE touching the tile walk.
# L2 (TCC) hits / misses / requests and the fabric-side read requests with their stall counters, per dispatch, on tools/microbench/gemm_shapes.py
# (enc gate|up, enc qkv, ... one launch shape per row). Each --pmc set in its own run with --kernel-trace only.
#   gpurun --timeout 900 -- 'bash tools/profile_bigtile_pmc.sh r06'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
CMD="python $R/tools/microbench/gemm_shapes.py"
for shape in "enc gate|up" "enc qkv"; do
export ONLY_SHAPE="$shape"
sn=$(echo "$shape" | tr -c 'a-z\n' '_')
i=1
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCC_TAG_STALL_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format rocpd -d /tmp/bp_${sn}_$i -- $CMD > /tmp/bp$i.out 2> /tmp/bp$i.err
  i=$((i + 1))
done
for k in 1 2 3 4; do
  d=$(find /tmp/bp_${sn}_$k -name "*.db" | head -1)
  if [ -n "$d" ]; then python $R/tools/rocpd_pmc.py --raw $d | grep -E "kernel|---|gemm_nt" >> $R/gpurun_out/${TAG}_bigtile_pmc_${sn}.md 2>&1; else tail -5 /tmp/bp$k.err >> $R/gpurun_out/${TAG}_bigtile_pmc_${sn}.md; fi
done
cat /tmp/bp1.out | head -3
done
grep -c "" $R/gpurun_out/${TAG}_bigtile_pmc_*.md
