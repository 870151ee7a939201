for r in 1 2 3; do
for v in "" "FFGPU_DW_XCD=1" "FFGPU_DW_BAND=8 FFGPU_DW_XCD=1"; do
  echo "[$v] $(env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-node-line 2>/dev/null | python -c '
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline_pw"]["us_per_launch"])')"
done; done
