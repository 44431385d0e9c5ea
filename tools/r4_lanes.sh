#!/bin/bash
# lanes x frames per launch, the main timed loop only (200 steps and the driver's 20)
for lanes in 2 3 4 5; do for pair in 6 8 9; do
  for steps in 200 20; do
    timeout 200 python bench.py --lanes $lanes --pair $pair --steps $steps --warmup 5 --main-only --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err
    python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('lanes $lanes pair $pair steps $steps', round(d['value'],1))" || tail -2 /tmp/b.err
  done
done; done
