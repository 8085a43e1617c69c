#!/bin/bash
for e in "$@"; do
  echo "== $e"
  env $e timeout 200 python bench.py --workload stories15M --steps 255 --no-cpu-baseline --no-extra 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']['by_kind']
        print('tok/s %.1f us %.1f |'%(d['value'],d['ms_per_step']*1e3),' '.join('%s %.1fus'%(k,v['ms_per_launch']*1e3) for k,v in r.items()))
    elif 'rror' in l: print(l.strip())
"
done
