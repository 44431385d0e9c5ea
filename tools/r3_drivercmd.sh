#!/bin/bash
# the driver's own command, timed end to end: gpurun -- 'bash tools/r3_drivercmd.sh'
mkdir -p gpurun_out/r3quick
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3quick/driver_cmd.json 2> gpurun_out/r3quick/driver_cmd.err
echo "elapsed $(echo "$(date +%s.%N) - $S" | bc) s"; tail -2 gpurun_out/r3quick/driver_cmd.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3quick/driver_cmd.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["config"]["frames_per_launch"], d["config"]["frames_in_flight"], "frac", r["frac"], "path", r["path_frac"], "traffic", r["traffic"], r["traffic_source"])
print("radar", {k: (v if not isinstance(v, str) else v[:40]) for k, v in r["others"]["chain_kernel(radar fusion)"].items()})
print(d["latency_ms_per_frame"], d["dropin_forward"]["channels_last"]["ms_per_frame"], d["pipeline_latency_ms"]["p50"], d["end_to_end"]["head_share_ms"], d["with_handoff"]["value"])
print(sorted(d.keys()))
PY
