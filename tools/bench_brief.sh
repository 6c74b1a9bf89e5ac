#!/bin/bash
# one-line summary of a bench.py run: value, ms per step, average launch ms, parity block   usage: tools/bench_brief.sh [bench.py args]
python bench.py "$@" 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%.0f it/s  ms/step %.1f  avg launch ms %.1f  frac %.4f  kernel %s  parity %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('kernel'), d.get('parity')))
"
