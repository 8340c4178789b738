cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pf -- python $R/tools/microbench/decode_sweep.py --configs fp8only --steps 16 > /tmp/o1 2>/tmp/e1
cd $R
python tools/rocpd_stats.py $(find /tmp/pf -name "*.db" | head -1) --by-grid > gpurun_out/fp8_decode_by_grid.md 2>&1
tail -3 /tmp/o1; head -40 gpurun_out/fp8_decode_by_grid.md | cut -c1-200
