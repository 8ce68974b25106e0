python3 bench.py --steps 20 --warmup 3 --cpu-sites 0 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['lnl'], d['kernels'])"
python3 bench.py --steps 20 --warmup 3 --cpu-sites 0 --tip-clv --taxa 16 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['lnl'], d['kernels'])"
