#!/bin/bash
python bench.py --no-cpu --stream-batch 0 --steps 40 --grants-mix | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grants-mix', d['value'], d['ms_per_step'])"
python bench.py --no-cpu --stream-batch 0 --steps 40 --grants | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grants', d['value'], d['ms_per_step'])"
