python3 bench.py --steps 20 --warmup 3 --cpu-sites 0 --sites 500000 --taxa 32 --newton 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['lnl'], d['kernels'], d['newton'])"
python3 tools/call_latency.py 2>&1 | tail -6
