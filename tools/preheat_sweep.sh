# bench.py's 20-step window after different pre-heat durations (two runs each)
for p in 0 20 60 150 400; do for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-agc-variant --preheat-ms $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('preheat ms', d['config']['preheat_steps'], 'steps:', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
done; done
