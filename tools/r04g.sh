# round 4, session 3: streamed detect -> recognise call, two-buffer decode attention. gpurun --timeout 1500 -- 'bash tools/r04g.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_attn_ops.py tests/test_gpu_round4.py tests/test_gpu_predictors.py -x -q -m gpu > $O/r04g_tests.txt 2>&1
tail -5 $O/r04g_tests.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/microbench/decode_attn2_bench.hip -o /tmp/da2 2>&1 | tail -3
: > $O/r04g_decode_attn_db.txt
for slots in 128 256; do for ctx in 140 200 300 590 900; do for v in 4 5; do
  timeout 120 /tmp/da2 $v $ctx 10 1 $slots 2>&1 | grep -v "launch status" | tr '\n' ' ' >> $O/r04g_decode_attn_db.txt; echo " slots=$slots" >> $O/r04g_decode_attn_db.txt
done; done; done
cat $O/r04g_decode_attn_db.txt
timeout 600 python bench.py --e2e-only > $O/r04g_e2e.json 2> $O/r04g_e2e.err; tail -c 1500 $O/r04g_e2e.json
timeout 600 python bench.py --texify-only --tuning dattn_db=-1 > $O/r04g_texify_db_off.json 2> $O/r04g_texify_off.err
timeout 600 python bench.py --texify-only > $O/r04g_texify_db_on.json 2> $O/r04g_texify_on.err
python - <<'PY'
import json
for f in ("off", "on"):
    try:
        d = json.load(open(f"gpurun_out/r04g_texify_db_{f}.json"))
        print(f, {k: (d[k] if not isinstance(d[k], dict) else d[k].get("ms")) for k in ("ms", "fp8_decode", "fp8_decode_fp8_kv", "bf16_decode_fp8_kv")})
    except Exception as e:
        print(f, "failed", e)
PY
