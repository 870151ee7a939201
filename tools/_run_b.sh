for i in 1 2 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run $i: value %.0f  ms/step %.4f  dw frac %.4f  pw us %.1f  two_steps %.0f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_pw']['us_per_launch'], d['config']['two_steps_per_launch']['value']))"
done
