"""Data-parallel training with REAL kernels on more than one rank: two processes share the test box's
one GPU (gloo backend -- RCCL refuses two ranks on one device; on a multi-GPU node the same code runs one rank per
GPU over RCCL).  Every rank trains the fusion head on its own frames (fused row-chain forward / backward, device loss,
decoder prefetch); the flat gradient bucket is all-reduced once per iteration.  Reference: tools/dist_train.sh:7-9."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_end_with_the_same_parameters():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'multirank_worker.py')], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    except subprocess.TimeoutExpired:
        tails = []
        for p in procs:                      # never leave a rank behind on the GPU
            p.kill()
            o, e = p.communicate()
            tails.append((o or '')[-800:] + '\n' + (e or '')[-1500:])
        pytest.fail('a rank did not finish within 240 s:\n' + '\n-----\n'.join(tails))
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith('MULTIRANK ')]
    assert len(line) == 1, outs[0][0]
    parts = json.loads(line[0][len('MULTIRANK '):])
    a, b = parts
    assert a['finite'] and b['finite']
    # same parameters on both ranks after three all-reduced steps (bit-identical: one summed bucket, one AdamW)
    assert a['sum'] == b['sum'] and a['abs'] == b['abs'] and a['first'] == b['first']
    # different frames and different dropout streams (the rank is part of the seed): different local losses
    assert a['losses'] != b['losses'] and a['seed'] != b['seed']
    # Round 6: 1 + 4 collectives per iteration -- the loss normalisers (a few floats, HEAD:885-902) and
    # the gradient bucket in its four exchange chunks (fusion layer 3, 2, 1, radar encoders: together the whole bucket),
    # each started asynchronously right behind the launch that finished its weight gradients: when chunk g's collective
    # was issued, exactly g + 1 chunk launches of the iteration had been enqueued -- i.e. it is on its way before the
    # later weight-gradient launches (and the last one in particular) have even started
    for part in (a, b):
        assert len(part['calls']) == 5 * 3, part['calls']
        assert len(part['chunks']) == 4 and sum(part['chunks']) > 2_000_000 and part['chunk_order_ok']
        for it in range(3):
            small, chunks = part['calls'][5 * it], part['calls'][5 * it + 1:5 * it + 5]
            assert small[0] <= 8
            assert [c[0] for c in chunks] == part['chunks'] and all(c[1] is True for c in chunks), chunks
            assert [c[2] for c in chunks] == [4 * it + g + 1 for g in range(4)], chunks


def test_training_bench_with_two_ranks_finishes_and_reports_its_roofline():
    """`bench.py --train --gpus 2` (the ranks share the one GPU, gloo): the launcher starts both ranks, every
    collective of the run -- the bucket all-reduce, the loss normalisers inside the roofline's extra iteration --
    is entered by BOTH ranks (round 3: rank 0 alone ran the roofline and the job hung), rank 0 prints one line."""
    import signal
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--train', '--gpus', '2', '--share-gpu', '--backend', 'gloo',
           '--steps', '6', '--warmup', '2', '--min-window-s', '0.05', '--warmup-s', '0.05']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                # the launcher and its ranks: the process group started here
        out, err = p.communicate()
        pytest.fail('bench.py --train --gpus 2 did not finish within 240 s:\n' + (err or '')[-1500:])
    assert p.returncode == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-1000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2
    assert d['exposed_collective_ms']['iterations'] == 20 and 0.0 <= d['exposed_collective_ms']['median'] < 50.0
    assert d['roofline'] and d['roofline']['parts'] and len(d['per_rank']['frames_per_s']) == 2
