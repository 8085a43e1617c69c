#!/bin/bash
# usage: scripts/tune.sh "<env assignments>" ...   -> tokens/s of the 7B shape per setting
for e in "$@"; do
  echo "== $e"
  env $e timeout 200 python bench.py --steps ${STEPS:-255} --no-cpu-baseline --no-extra 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']['by_kind']
        print('tok/s %.1f ms %.3f |'%(d['value'],d['ms_per_step']),' '.join('%s %.1fus %s'%(k,v['ms_per_launch']*1e3, ('%.2fTB/s'%(v['GBps']/1e3)) if v['GBps'] else '') for k,v in r.items()))
    elif 'rror' in l: print(l.strip())
"
done
