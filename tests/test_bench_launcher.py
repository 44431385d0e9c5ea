"""`bench.py --gpus N` must really start N ranks (round-1 VERDICT: the flag was parsed and
ignored).  CPU dry run: the launcher parent spawns N fresh children, they rendezvous over gloo,
the rank census all-reduce sees N ranks, rank 0 prints ONE JSON line with n_gpus = N.  The same
for the training mode (one all-reduce of the real 10 MB flat gradient bucket per step) and under
the driver's own launcher (torch.distributed.run).  Reference: tools/dist_train.sh:7-9."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ['--dry-run', '--steps', '3', '--warmup', '1', '--min-window-s', '0.02', '--warmup-s', '0']


def _run(cmd, env=None):
    e = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize('mode', [[], ['--train']])
def test_gpus_flag_spawns_that_many_ranks(mode):
    line = _run([sys.executable, 'bench.py', '--gpus', '2'] + FAST + mode)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2
    assert line['config']['launcher'] == 'bench.py'
    assert line['steps'] == 3 and line['warmup'] == 1
    assert line['bucket_all_reduce_ok'] is True
    if mode:
        assert line['config']['grad_bucket_bytes'] > 9_000_000     # 2.5 M trainable fp32 values


def test_single_process_default():
    line = _run([sys.executable, 'bench.py'] + FAST)
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1
    assert line['config']['launcher'] == 'single process'


def test_under_torch_distributed_run():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    line = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                 '--master-addr', '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '2']
                + FAST)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2
    assert line['config']['launcher'] == 'torch.distributed.run'


def test_a_dead_rank_fails_the_launch():
    """a child that exits non-zero takes the launch down instead of hanging the others"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('on a GPU box the ranks would run the real bench')
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    # no GPU in this container and no --dry-run: every rank asserts
    assert r.returncode != 0
    assert 'needs MI355X' in r.stderr


def test_automatic_frames_per_launch_keeps_every_tile_resident():
    """--pair 0: the most frames whose 16-row tiles are resident at once, two workgroups per CU -- 900 queries on
    the 256 CUs of an MI355X: 9 frames = 507 workgroups (10 would be 563: a second scheduling round).  The window
    length does not enter: its remainder is one partial launch."""
    from transcar_amd.pipeline import resident_frames_per_launch as rfl
    assert rfl(900, num_cus=256) == 9
    assert -(-9 * 900 // 16) <= 512 < -(-10 * 900 // 16)
    assert rfl(300, num_cus=256) == 27
    assert rfl(10000, num_cus=256) == 1
    for q in range(100, 3000, 37):
        p = rfl(q, num_cus=256)
        assert p >= 1 and (p == 1 or -(-p * q // 16) <= 512)


CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'per_rank', 'cpu_affinity',
            'rccl_ranks', 'timing')


def test_every_world_size_prints_one_schema():
    """The N = 1 point of the driver's scaling run and the plain bench line must be comparable: the single
    process, `--gpus 2` (bench.py's own launcher) and torch.distributed.run all print the contract's keys plus
    `per_rank` (frames/s of each rank from its own windows: min / median / max) and `cpu_affinity`."""
    one = _run([sys.executable, 'bench.py'] + FAST)
    two = _run([sys.executable, 'bench.py', '--gpus', '2'] + FAST)
    for line in (one, two):
        for k in CONTRACT:
            assert k in line, k
    assert set(one) == set(two)
    assert len(one['per_rank']['frames_per_s']) == 1 and len(two['per_rank']['frames_per_s']) == 2
    assert two['per_rank']['min'] <= two['per_rank']['median'] <= two['per_rank']['max']
    assert one['cpu_affinity']['pinned'] is False            # a single rank keeps every core (cpu_baseline)


def test_rank_is_pinned_to_the_numa_node_of_its_gpu(tmp_path):
    """pin_to_gpu_numa_node reads /sys only (no GPU call: it runs before torch is imported): GPU i = the i-th AMD
    display / accelerator function in PCI order; its numa_node's cpulist becomes the rank's affinity mask."""
    import bench
    sysr = tmp_path / 'sys'
    mine = sorted(os.sched_getaffinity(0))
    half = max(1, len(mine) // 2)
    nodes = {0: mine[:half], 1: mine[half:] or mine[:half]}
    for n, cpus in nodes.items():
        d = sysr / 'devices' / 'system' / 'node' / ('node%d' % n)
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(','.join(str(c) for c in cpus) + '\n')
    pci = sysr / 'devices' / 'pci'
    for i, (addr, vendor, cls, node) in enumerate([('0000:05:00.0', '0x1002', '0x120000', 0),
                                                   ('0000:85:00.0', '0x1002', '0x120000', 1),
                                                   ('0000:03:00.0', '0x1a03', '0x030000', 0)]):      # a BMC VGA: skipped
        dev = pci / addr
        (dev / 'drm').mkdir(parents=True)
        (dev / 'vendor').write_text(vendor + '\n')
        (dev / 'class').write_text(cls + '\n')
        (dev / 'numa_node').write_text('%d\n' % node)
        card = sysr / 'class' / 'drm' / ('card%d' % i)
        card.mkdir(parents=True)
        (card / 'device').symlink_to(dev, target_is_directory=True)
    assert bench.gpu_numa_topology(str(sysr)) == [('0000:05:00.0', 0), ('0000:85:00.0', 1)]
    assert bench.parse_cpulist('0-3,8,10-11') == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    try:
        info = bench.pin_to_gpu_numa_node(1, str(sysr))
        assert info['pinned'] and info['numa_node'] == 1 and info['pci'] == '0000:85:00.0'
        assert os.sched_getaffinity(0) == set(nodes[1])
    finally:
        os.sched_setaffinity(0, before)
    assert bench.pin_to_gpu_numa_node(5, str(sysr))['pinned'] is False      # no such GPU: left alone
    # the ROCm runtime's own enumeration (KFD topology nodes) wins over PCI order: here node 2 = 85:00.0 comes first
    for nid, (simd, loc) in enumerate([(0, 0), (256, 0x8500), (256, 0x0500)]):
        d = sysr / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes' / str(nid)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count 64\nsimd_count %d\ndomain 0\nlocation_id %d\n' % (simd, loc))
    assert bench.kfd_gpu_order(str(sysr)) == ['0000:85:00.0', '0000:05:00.0']
    assert bench.gpu_numa_topology(str(sysr)) == [('0000:85:00.0', 1), ('0000:05:00.0', 0)]
    old = {k: os.environ.pop(k, None) for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')}
    try:
        os.environ['HIP_VISIBLE_DEVICES'] = '1'
        assert bench.gpu_numa_topology(str(sysr)) == [('0000:05:00.0', 0)]
    finally:
        os.environ.pop('HIP_VISIBLE_DEVICES', None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def test_pmc_counter_csv_is_reduced_per_kernel(tmp_path):
    """bench.py's own PMC passes (roofline.traffic / mfma_busy measured in the run): the rocprofv3 counter csv is
    averaged per launch over the path's kernels; the dual launches (decoder layer + radar encoder half) and other
    counters' rows are left out."""
    import bench
    f = tmp_path / 'x_counter_collection.csv'
    kd = 'void tc::(anonymous namespace)::chain_kernel<16, 1, false>(tc::(anonymous namespace)::ChainDev, ...)'
    kr = 'void tc::(anonymous namespace)::chain_kernel<16, 3, false>(tc::(anonymous namespace)::ChainDev, ...)'
    ku = 'void tc::(anonymous namespace)::chain_dual_kernel<16, 16, 4>(...)'
    ka = 'void tc::self_attn_kernel<2, false>(float const*, ...)'
    rows = [(kd, 'FETCH_SIZE', 100.0, 0, 1000), (kd, 'FETCH_SIZE', 300.0, 2000, 4000), (kd, 'WRITE_SIZE', 7.0, 0, 10),
            (kr, 'FETCH_SIZE', 50.0, 0, 500), (ku, 'FETCH_SIZE', 999.0, 0, 1), (ka, 'FETCH_SIZE', 10.0, 0, 2000)]
    f.write_text('Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n' +
                 ''.join('"%s",%s,%s,%d,%d\n' % r for r in rows))
    got = bench.parse_counter_csv(str(f), 'FETCH_SIZE')
    assert set(got) == {'chain_kernel(decoder layer)', 'chain_kernel(radar fusion)', 'self_attn_kernel'}
    v, n, dur = got['chain_kernel(decoder layer)']
    assert v == 200.0 and n == 2 and abs(dur - 1.5e-6) < 1e-12
    assert got['chain_kernel(radar fusion)'][0] == 50.0 and got['self_attn_kernel'][1] == 1
    assert bench.parse_counter_csv(str(f), 'WRITE_SIZE') == {'chain_kernel(decoder layer)': (7.0, 1, 1e-8)}


def test_training_iteration_traffic_is_summed_over_all_dispatches(tmp_path):
    """`bench.py --train` -> roofline.traffic: every dispatch of the child's counter csv counts, the number of
    iterations is the number of adamw_kernel launches (one per iteration), bytes = (2 FETCH + WRITE) * 1024."""
    import bench
    hdr = 'Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n'
    names = ['void tc::chain_kernel<4, 6, true>(...)', 'void tc::bwd_weight_group_kernel(...)', 'void tc::adamw_kernel(...)']
    fetch, write = tmp_path / 'f.csv', tmp_path / 'w.csv'
    fetch.write_text(hdr + ''.join('"%s",FETCH_SIZE,%s,0,1\n' % (n, v) for _ in range(3) for n, v in zip(names, (100., 40., 10.))))
    write.write_text(hdr + ''.join('"%s",WRITE_SIZE,%s,0,1\n' % (n, v) for _ in range(3) for n, v in zip(names, (20., 30., 10.))) +
                     '"x",FETCH_SIZE,999,0,1\n')
    got = bench.iteration_traffic_from_csvs(str(fetch), str(write))
    assert got['iterations'] == 3 and got['fetch_kb'] == 150.0 and got['write_kb'] == 60.0
    assert got['traffic_bytes'] == (2 * 150 + 60) * 1024
    write.write_text(hdr)
    assert bench.iteration_traffic_from_csvs(str(fetch), str(write)) is None


def test_cpu_baseline_spread_reads_the_committed_lines():
    """cpu_baseline.box_to_box: min / max / median of cpu_baseline.ms_per_frame over the committed
    bench lines of every round plus the current run's figure."""
    import bench
    got = bench.cpu_baseline_spread(1000.0)
    assert got['runs'] >= 2 and got['max_ms_per_frame'] == 1000.0
    assert 100.0 < got['min_ms_per_frame'] <= got['median_ms_per_frame'] <= got['max_ms_per_frame']
    base = bench.cpu_baseline_spread()
    assert base['runs'] == got['runs'] - 1 and base['max_ms_per_frame'] < 1000.0
