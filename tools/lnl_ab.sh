python3 bench.py --steps 20 --warmup 3 --cpu-sites 0 --states 20 --sites 200000 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['lnl'], d['roofline']['avg_op_us'], d['kernels'])"
