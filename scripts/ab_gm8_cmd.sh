#!/bin/bash
python bench.py --no-cpu --stream-batch 0 --steps 40 --grants-mix --llr8 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grants-mix llr8', d['value'], d['ms_per_step'])"
