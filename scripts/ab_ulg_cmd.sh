#!/bin/bash
python scripts/bench_ul_grants.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['subframes_per_s'], v['ms_per_batch']) for k,v in d.items() if k[0]!='_'})"
