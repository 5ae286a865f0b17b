export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'frac', d['roofline']['frac']); print('from_q', {k: v for k, v in d['from_q'].items() if k != 'note'})"
