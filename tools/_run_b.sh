python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/b20.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b20.json').read())
print(d['value'], d['config'].get('fp32_resident_input',{}).get('value'), d['strong_b256']['merged']['value'], d['strong_b256']['unmerged']['value'], d['c_node_api']['value'], d['config']['two_steps_per_launch']['value'])
PY
