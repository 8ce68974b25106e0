for i in 1 2 3 4 5; do
python3 bench.py --cpu-sites 0 --no-vary --no-c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', d['value'], d['ms_per_step'], d['roofline']['frac'], d['ramp'], d['api_calls']['update_partials_ms_hip_events'])"
done
