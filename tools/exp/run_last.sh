mkdir -p gpurun_out/r06t
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(time python bench.py) > gpurun_out/r06t/bench_line_default.json 2> gpurun_out/r06t/bench_default_stderr.log
tail -4 gpurun_out/r06t/bench_default_stderr.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06t/bench_line_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline'])
print([ (c['config'][:30], c.get('device_resident_ms'), c.get('create_ms')) for c in d['configs']])
PY
