"""The driver's own command, end to end on the GPU box (runs LAST: the file name sorts behind the parity tests):
`python bench.py --gpus 1 --steps 20 --warmup 5` prints exactly one JSON line with the contract's keys, the roofline
of the dominant kernel with its HBM-side traffic (measured in the run when rocprofv3 is there) and the CPU baseline."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def test_the_drivers_bench_command_prints_one_complete_line():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5'],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)            # the process group started here (bench.py and its rocprofv3 passes)
        out, err = p.communicate()
        pytest.fail('bench.py did not finish within 420 s:\n' + (err or '')[-1500:])
    assert p.returncode == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-1000:]
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['higher_is_better'] is True
    assert d['unit'] == 'frames/s' and d['value'] > 1000 and d['vs_baseline'] is None
    # fp32 results: the linear steps of the batched launches take two-plane f16 operands on the matrix cores
    assert d['dtype'].startswith('f32') and 'f16x2' in d['dtype']
    assert abs(d['ms_per_step'] * d['value'] - 1e3) < 1.0                  # one frame per step
    assert 'configs[1]' in d['config']['workload'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    # priced against the f16 matrix-core peak / 3 products (838.9 TFLOP/s); the f32-MFMA (vector pipe) figure beside it
    assert r['peak'] > 800 and 'peak_definition' in r and 0.05 < r['frac'] < 1.0 and r['frac_of_f32_mfma_peak'] > 0.5
    assert r['traffic'] and r['traffic'] > 1e8                              # ~200 MB per launch of nine frames
    # the plain-f32 path, configs[4] (VoVNet levels) and configs[2] (a training iteration) ride in the same line
    assert d['f32_path']['matrix_path'] == 'f32' and 1000 < d['f32_path']['value'] < d['value'] * 1.05
    assert 'configs[4]' in d['vovnet']['workload'] and d['vovnet']['value'] > 1000
    assert 'configs[2]' in d['train']['workload'] and 0.1 < d['train']['ms_per_iteration'] < 5.0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['sample']
