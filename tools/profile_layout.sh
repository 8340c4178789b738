# Per-kernel times of the layout leg (bench.py --layout-only) and of the texify leg at a given horizon: one kernel-trace pass each.
#   gpurun --timeout 500 -- 'bash tools/profile_layout.sh r03f 768'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-lay}
T=${2:-768}
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pl1 -- python $R/bench.py --layout-only --no-cpu-baseline > $R/gpurun_out/${TAG}_layout_line.json 2>/tmp/el1
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pl2 -- python $R/bench.py --texify-only --texify-tokens $T > $R/gpurun_out/${TAG}_texify${T}_line.json 2>/tmp/el2
cd $R
python tools/rocpd_stats.py $(find /tmp/pl1 -name "*.db" | head -1) --by-grid > gpurun_out/${TAG}_layout_kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find /tmp/pl2 -name "*.db" | head -1) > gpurun_out/${TAG}_texify${T}_kernel_stats.md 2>&1
cat gpurun_out/${TAG}_layout_line.json | cut -c1-600
head -45 gpurun_out/${TAG}_layout_kernel_stats.md | cut -c1-210
cat gpurun_out/${TAG}_texify${T}_line.json | cut -c1-1200
head -30 gpurun_out/${TAG}_texify${T}_kernel_stats.md | cut -c1-210
tail -3 /tmp/el1 /tmp/el2
