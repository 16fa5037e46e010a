# -*- coding: utf-8 -*-
"""The multi-rank code paths on ONE GPU: a world of one rank with the collectives forced on
(KGE_FORCE_COLLECTIVES=1, RCCL backend) must reproduce the single-GPU metrics -- entity shards
exchanging counts (hipGraph segments with the all-reduces between them) and query shards."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, forced, port):
    env = dict(os.environ)
    env.pop('KGE_FORCE_COLLECTIVES', None)
    if forced:
        env.update({'KGE_FORCE_COLLECTIVES': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1',
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--workload', 'complex_wn18rr', '--no-cpu-baseline', '--no-secondary'] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert lines and out.stdout.strip().splitlines()[-1] == lines[-1]      # the JSON line is the last line of stdout
    return json.loads(lines[-1])


def test_forced_collectives_reproduce_single_gpu_metrics():
    ref = _bench([], False, 0)
    assert ref['config']['parallelism'] == 'single'
    for k, extra in enumerate((['--scaling', 'strong', '--shard', 'entities', '--exchange', 'counts'],
                               ['--scaling', 'strong', '--shard', 'queries'])):
        got = _bench(extra, True, 29731 + k)
        assert got['config']['parallelism'] != 'single' and got['config']['hip_graph'] is True
        assert got['filtered_mrr'] == ref['filtered_mrr']
        assert got['filtered_hits_at_10'] == ref['filtered_hits_at_10']
