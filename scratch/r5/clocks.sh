#!/bin/bash
# engine clock and power while the headline stream runs (is the chip clock- or power-limited under this load?)
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu --no-variants --no-harvest --no-aperiodic --no-configs --steps 400 --warmup 20 --reps 3 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 14
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g; s/  */ /g'; echo; sleep 0.7; done
wait $BP
python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('value', d['value'], d['ms_per_step'])"
