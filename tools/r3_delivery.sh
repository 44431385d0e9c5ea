for cfg in "4 3" "8 4" "8 5"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --lanes $2 --steps 180 --warmup 10 --no-cpu-baseline --no-batched --no-handoff --no-roofline 2>/dev/null > /tmp/d.json
  python -c "import json; d=json.loads(open('/tmp/d.json').read().strip().splitlines()[-1]); print('queues $1 lanes $2:', round(d['value']), 'delivery', round(d['with_input_delivery']['tokens_l2i_h2d']['value']), 'pipe p50', round(d['pipeline_latency_ms']['p50'],2))"
done
