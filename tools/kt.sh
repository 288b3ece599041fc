python bench.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d.get('parity'))"
