mkdir -p gpurun_out/r06t
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06t/pytest_gpu.log 2>&1
tail -4 gpurun_out/r06t/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06t/bench_line.json 2> gpurun_out/r06t/bench_stderr.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06t/bench_line.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
for c in d['configs']:
    for k in c:
        if 'create_ms' in str(c[k]) or k=='create_ms': print(k, str(c[k])[:400])
print(json.dumps(d['batch'].get('one_old_many_new'))[:500])
PY
grep -o '"create_ms[^}]*}' gpurun_out/r06t/bench_line.json | head -3
