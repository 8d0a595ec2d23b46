"""bench.py contract: one JSON line with the driver's keys, the roofline / cpu_baseline objects, and kernel names that the
library really launched (so that the per-kernel table can be laid beside rocprofv3's).  Runs the toy configuration."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'toy', '--steps', '2', '--warmup', '1'] + list(extra),
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run('--no-fp32-leg')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['value'] > 0 and 'workload' in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
        assert k in r, k
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    names = list(r['all_conv_kernels']['by_kernel'])
    assert names and all(n.startswith('conv_') and 'kernel' in n for n in names), names      # HIP kernel names, not labels
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['value'] > 0


def test_library_reports_the_launched_kernel(device):
    import torch
    from ag2video_amd import _lib, ops
    x = torch.randn(2, 16, 16, 16, device=device)
    w = torch.randn(160, 16, 3, 3, device=device)
    ops.conv2d(x, w, None, 1, 1)
    assert b'conv_igemm_bf16x3_ws_kernel' in _lib.query('ag2v_last_conv_kernel')
    # the row threshold travels with the call (ag2v_conv_opts::min_rows): the host never consults the library's per-device default
    old = _lib.query('ag2v_get_bf16x3_min_rows')
    try:
        _lib.call('ag2v_set_bf16x3_min_rows', 512)      # deprecated shim: edits the current device's default only ...
        ops.conv2d(x, w, None, 1, 1)
        assert b'conv_igemm_bf16x3_ws_kernel' in _lib.query('ag2v_last_conv_kernel')                # ... which ops.py does not read
        with pytest.raises(RuntimeError):
            _lib.call('ag2v_set_bf16x3_min_rows', 0)
    finally:
        _lib.call('ag2v_set_bf16x3_min_rows', old)
    with ops.bf16x3_min_rows(512):
        ops.conv2d(x, w, None, 1, 1)
        assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_igemm_f32_ws_kernel<128')      # exact arithmetic, >= 128 rows
        ops.conv2d(x, w[:96], None, 1, 1)
        assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_igemm_kernel<128')             # ... fewer rows


def test_build_then_smoke_in_one_process():
    """The driver's two entry points in one interpreter: build() loads the library before anything imported torch,
    which must not leave the process on a HIP runtime torch was not built against."""
    out = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke()'],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'smoke ok' in out.stdout


def test_bench_launches_its_own_ranks_over_rccl():
    """`bench.py --gpus N` without a torchrun environment starts the N ranks itself (a child `torch.distributed.run`) and relays
    rank 0's line; with AG2V_FORCE_DIST=1 the same launcher path runs at N = 1, so the RCCL process group is real on the test box."""
    env = dict(os.environ, AG2V_FORCE_DIST='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'toy', '--gpus', '1', '--steps', '2', '--warmup', '1',
                          '--no-fp32-leg', '--no-cpu-baseline'], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['comm']['rccl_world'] == 1 and d['comm']['backend'] == 'nccl', d['comm']
    assert 'g_allreduce_ms_alone' in d['comm']
