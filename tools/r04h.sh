# round 4, session 3: layout encoder LayerNorm / self-attention kernel A/B, detection with the persistent tile loop. gpurun --timeout 1500 -- 'bash tools/r04h.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_layout.py tests/test_gpu_layout_fed.py tests/test_gpu_table.py tests/test_gpu_predictors.py tests/test_gpu_round4.py -x -q -m gpu > $O/r04h_tests.txt 2>&1
tail -5 $O/r04h_tests.txt
timeout 400 python bench.py --layout-only --no-cpu-baseline --tuning lay_ln=0 --tuning dattn=3 > $O/r04h_layout_before.json 2> $O/r04h_layout_before.err
timeout 400 python bench.py --layout-only --no-cpu-baseline > $O/r04h_layout_after.json 2> $O/r04h_layout_after.err
python - <<'PY'
import json
for f in ("before", "after"):
    try:
        d = json.load(open(f"gpurun_out/r04h_layout_{f}.json"))
        print(f, {k: {kk: d[k].get(kk) for kk in ("pages_per_s", "tables_per_s", "encode_ms", "decode_step_us", "batch_128")} for k in d})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 400 python tools/microbench/det_sweep.py persist=1 > $O/r04h_det_sweep.txt 2>&1; grep -v amdgpu.ids $O/r04h_det_sweep.txt
