# round 4, session 3: folded decode head of the detector. gpurun --timeout 1200 -- 'bash tools/r04k.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_det.py tests/test_gpu_baseline_parity.py tests/test_gpu_det_post.py -x -q -m gpu -s > $O/r04k_tests.txt 2>&1
grep -n "folded vs unfolded\|passed\|failed\|Error" $O/r04k_tests.txt | tail -12
DETECTOR_HEAD_UNFOLDED=1 timeout 300 python bench.py --det-only --no-cpu-baseline --det-steps 10 > $O/r04k_det_unfolded.json 2> $O/r04k_det_unfolded.err
timeout 300 python bench.py --det-only --no-cpu-baseline --det-steps 10 > $O/r04k_det_folded.json 2> $O/r04k_det_folded.err
timeout 300 python bench.py --det-only --no-cpu-baseline --det-steps 10 --tuning det_head_blk=2 > $O/r04k_det_unfolded2.json 2>> $O/r04k_det_unfolded.err
timeout 300 python bench.py --det-only --no-cpu-baseline --det-steps 10 > $O/r04k_det_folded2.json 2>> $O/r04k_det_folded.err
python - <<'PY'
import json
for f in ("unfolded", "folded", "unfolded2", "folded2"):
    try:
        d = json.load(open(f"gpurun_out/r04k_det_{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["parity"], d["roofline"]["achieved"])
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python $GRAFT_REPO_ROOT/bench.py --det-only --no-cpu-baseline --det-steps 2 > /tmp/ok 2>/tmp/ek
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $(find /tmp/pk -name "*.db" | head -1) > $O/r04k_det_kernel_stats.md 2>&1; head -30 $O/r04k_det_kernel_stats.md | cut -c1-160
