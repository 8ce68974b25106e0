python3 bench.py --steps 10 --warmup 2 --cpu-sites 0 --sites 500000 --taxa 200 --tree random --newton 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['workload']); print(d['kernels']); print(d['roofline']); print(d['newton'])"
