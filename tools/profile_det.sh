# Per-kernel times of the detection forward (bench.py --det-only): one kernel-trace pass, statistics per kernel and per (kernel, grid).
#   gpurun --timeout 400 -- 'bash tools/profile_det.sh r03x'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-det}
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd1 -- python $R/bench.py --det-only --no-cpu-baseline --det-steps 5 > $R/gpurun_out/${TAG}_det_line.json 2>/tmp/ed1
cd $R
python tools/rocpd_stats.py $(find /tmp/pd1 -name "*.db" | head -1) --by-grid > gpurun_out/${TAG}_det_kernel_stats.md 2>&1
head -40 gpurun_out/${TAG}_det_kernel_stats.md | cut -c1-200
