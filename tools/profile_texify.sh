cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pt -- python $R/bench.py --no-cpu-baseline --no-det --no-e2e --steps 1 --warmup 0 > /tmp/o1 2>/tmp/e1
cd $R
python tools/rocpd_stats.py $(find /tmp/pt -name "*.db" | head -1) --by-grid > gpurun_out/texify_by_grid.md 2>&1
tail -c 600 /tmp/o1; grep -n "decode_attn_flash\|gemm_mx\|splitk_residual" gpurun_out/texify_by_grid.md | cut -c1-190
