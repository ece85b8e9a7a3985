"""Print the fields of a bench.py JSON line that matter when comparing variants (reads the line from stdin)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    c = d.get('config', {})
    r = d.get('roofline') or {}
    print('ms/step %.3f  value %.4g  stages %s  spmv %s kernel %s frac %s' % (
        d['ms_per_step'], d['value'] or 0, json.dumps(c.get('stage_ms')), r.get('avg_launch_ms'), r.get('kernel'), r.get('frac')))
