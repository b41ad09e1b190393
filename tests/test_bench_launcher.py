"""bench.py's own multi-rank launcher (CPU, gloo, world size 2): `python bench.py --gpus N` must start N ranks itself
(tools/dist_test.sh:9-10 launches one process per GPU for the reference), report n_gpus = N, the per-rank rates and the
world size the process group saw, take the MAX over ranks as the step time, and refuse loudly when fewer than N devices
are visible.  `--stub` replaces the GPU window by a fixed sleep: this exercises the launch / barrier / reduction / JSON
contract, not a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=300, env=env)


def test_gpus_2_launches_two_ranks_and_reports_the_slowest():
    r = _run(['--gpus', '2', '--stub', '--steps', '10', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE json line
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['gpus_requested'] == 2 and d['rccl_world_size'] == 2
    assert d['steps'] == 10 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['higher_is_better'] is True
    assert [p['rank'] for p in d['per_rank']] == [0, 1]
    # rank 1 sleeps twice as long per step: the step time is the slowest rank's (max over ranks), not rank 0's
    assert d['ms_per_step'] >= 4.0
    assert d['per_rank'][0]['frames_per_s'] >= d['per_rank'][1]['frames_per_s'] * 0.9
    slow = min(p['frames_per_s'] for p in d['per_rank'])
    assert abs(d['value'] - 2 * slow) <= 0.1 * d['value']     # whole-job value = N * steps / max time
    ar = d['train_allreduce']
    assert ar['backend'] == 'gloo' and ar['ms'] > 0 and ar['bytes'] > 0 and ar['algbw_gbs'] > 0


def test_single_rank_needs_no_launcher():
    r = _run(['--stub', '--steps', '3', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['rccl_world_size'] == 1 and 'train_allreduce' not in d


def test_refuses_more_ranks_than_visible_gpus():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(['--gpus', str(have + 2), '--no-cpu-baseline', '--no-train-step'])
    assert r.returncode == 2
    assert 'refusing' in r.stderr and ('--gpus %d' % (have + 2)) in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
