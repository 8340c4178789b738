# (SKIP_REC_PMC=1: skip the two recognition counter passes -- for calls after a change that touched detection kernels only.)
# End-of-round evidence: kernel trace of the default bench command (short) + the FETCH_SIZE / WRITE_SIZE counter passes behind
# profiles/hbm_traffic.json.   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r03k'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03k}
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5     # a box can be handed out twice in a session: `find | head -1` below must not pick up the previous call's database
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p1 -- python $R/bench.py --no-cpu-baseline --no-texify --no-layout --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/tmp/e1
[ -n "$SKIP_REC_PMC" ] || rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/p2 -- python $R/bench.py --no-cpu-baseline --no-det --no-e2e --no-texify --no-layout --steps 1 --warmup 0 > /tmp/o2 2>/tmp/e2
[ -n "$SKIP_REC_PMC" ] || rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/p3 -- python $R/bench.py --no-cpu-baseline --no-det --no-e2e --no-texify --no-layout --steps 1 --warmup 0 > /tmp/o3 2>/tmp/e3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/p4 -- python $R/bench.py --det-only --no-cpu-baseline --no-det-op-list --det-steps 2 > /tmp/o4 2>/tmp/e4
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/p5 -- python $R/bench.py --det-only --no-cpu-baseline --no-det-op-list --det-steps 2 > /tmp/o5 2>/tmp/e5
cd $R
db() { find /tmp/$1 -name "*.db" | head -1; }
python tools/rocpd_stats.py $(db p1) --by-grid > gpurun_out/${TAG}_kernel_stats.md 2>&1
[ -n "$SKIP_REC_PMC" ] || python tools/rocpd_pmc.py $(db p2) $(db p3) > gpurun_out/${TAG}_hbm_traffic_pmc.md 2>&1
[ -n "$SKIP_REC_PMC" ] || python tools/rocpd_pmc.py --json $(db p2) $(db p3) > gpurun_out/${TAG}_hbm_traffic.json 2>&1
python tools/rocpd_pmc.py $(db p4) $(db p5) > gpurun_out/${TAG}_det_hbm_traffic_pmc.md 2>&1
python tools/rocpd_pmc.py --det-json $(db p4) $(db p5) > gpurun_out/${TAG}_det_hbm_traffic.json 2>&1
tail -2 /tmp/e1; cat gpurun_out/${TAG}_hbm_traffic.json gpurun_out/${TAG}_det_hbm_traffic.json 2>/dev/null; head -14 gpurun_out/${TAG}_kernel_stats.md | cut -c1-150
