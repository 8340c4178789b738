# round 4, session 3: parallel staging copies in the detector / line pre-processor. gpurun --timeout 1200 -- 'bash tools/r04l.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_predictors.py tests/test_gpu_resample.py tests/test_gpu_prep.py tests/test_gpu_edge_cases.py tests/test_gpu_det_post.py -x -q -m gpu > $O/r04l_tests.txt 2>&1
tail -3 $O/r04l_tests.txt
timeout 400 python bench.py --det-only --no-cpu-baseline --det-steps 10 > $O/r04l_det.json 2> $O/r04l_det.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04l_det.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d["e2e"], indent=0)[:900])
PY
timeout 400 python bench.py --e2e-only > $O/r04l_e2e.json 2> $O/r04l_e2e.err
python - <<'PY'
import json
e = json.load(open("gpurun_out/r04l_e2e.json"))
print(e["pages_per_s"], e["lines_per_s"], e["wall_ms"], e["wall_ms_all_passes"], e["recognise_phases_ms"], e["serial_schedule"])
PY
