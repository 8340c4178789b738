cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pe -- python $R/bench.py --no-cpu-baseline --no-texify --no-det --steps 1 --warmup 0 > /tmp/o1 2>/tmp/e1
cd $R
python tools/rocpd_stats.py $(find /tmp/pe -name "*.db" | head -1) > gpurun_out/e2e_kernel_stats.md 2>&1
grep -n "prep_\|u8_to\|post_" gpurun_out/e2e_kernel_stats.md | cut -c1-160; tail -c 700 /tmp/o1
