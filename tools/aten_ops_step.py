"""Where the ATen kernels of a c3 gd_step come from: torch.profiler with shapes and Python stacks, the copy / add / cat / fill
operators above a device-time threshold grouped by (operator, shapes, innermost repo frame)."""
import collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from ag2video_amd.engine import TrainStep
opt, side, frames, batch = bench.make_opt('c3')
ts = TrainStep(opt, 'cuda', weights_seed=0)
clip = ts.to_device(bench.make_clip(opt, 'c3', batch, frames, side, seed=1))
for _ in range(2):
    ts.gd_step(clip)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.gd_step(clip)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    dt = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if dt <= 0:
        continue
    tot += dt
    frame = ''
    for s in (ev.stack or []):
        if '/ag2video_amd/' in s or 'bench.py' in s or '/scripts/' in s:
            frame = s.split('/ag2video_amd/')[-1] if '/ag2video_amd/' in s else s
            break
    key = (ev.name, str(ev.input_shapes)[:90], frame[:90])
    rows[key][0] += 1
    rows[key][1] += dt
print('ATen device time in the step: %.2f ms' % (tot / 1e3))
for key, (n, dt) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%7.3f ms %4d x  %-22s %-90s %s' % (dt / 1e3, n, key[0], key[1], key[2]))

# torch.cat / torch.stack by call site (the profiler does not record the shapes of tensor-list inputs): output megabytes per step
import traceback
sites = collections.defaultdict(lambda: [0, 0.0])
_cat, _stack = torch.cat, torch.stack


def _wrap(fn, name):
    def f(*a, **k):
        out = fn(*a, **k)
        fr = [s for s in traceback.extract_stack()[:-1] if '/ag2video_amd/' in s.filename]
        where = '%s:%d' % (fr[-1].filename.split('/ag2video_amd/')[-1], fr[-1].lineno) if fr else '?'
        sites[(name, where)][0] += 1
        sites[(name, where)][1] += out.numel() * out.element_size() / 1e6
        return out
    return f


torch.cat, torch.stack = _wrap(_cat, 'cat'), _wrap(_stack, 'stack')
ts.gd_step(clip)
torch.cuda.synchronize()
torch.cat, torch.stack = _cat, _stack
for key, (n, mb) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%9.1f MB %4d x  %-6s %s' % (mb, n, key[0], key[1]))

# small fills / copies by call site (each is a ~7 us rocclr fill / copy kernel on the stream: ~770 per step)
small = collections.defaultdict(lambda: [0, 0])
_orig = {}


def _site():
    fr = [s for s in traceback.extract_stack()[:-2] if '/ag2video_amd/' in s.filename]
    return '%s:%d' % (fr[-1].filename.split('/ag2video_amd/')[-1], fr[-1].lineno) if fr else '?'


def _wrap_method(name):
    orig = getattr(torch.Tensor, name)
    _orig[name] = orig

    def f(self, *a, **k):
        if self.is_cuda:
            rec = small[(name, _site())]
            rec[0] += 1
            rec[1] += self.numel()
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)


def _wrap_fn(name):
    orig = getattr(torch, name)
    _orig['torch.' + name] = orig

    def f(*a, **k):
        out = orig(*a, **k)
        if torch.is_tensor(out) and out.is_cuda:
            rec = small[(name, _site())]
            rec[0] += 1
            rec[1] += out.numel()
        return out
    setattr(torch, name, f)


for nm in ('copy_', 'zero_', 'fill_'):
    _wrap_method(nm)
for nm in ('zeros', 'zeros_like', 'ones', 'full', 'tensor', 'clone'):
    _wrap_fn(nm)
ts.gd_step(clip)
torch.cuda.synchronize()
for nm in ('copy_', 'zero_', 'fill_'):
    setattr(torch.Tensor, nm, _orig[nm])
for nm in ('zeros', 'zeros_like', 'ones', 'full', 'tensor', 'clone'):
    setattr(torch, nm, _orig['torch.' + nm])
print('fills / copies per step by call site (calls, elements):')
for key, (n, el) in sorted(small.items(), key=lambda kv: -kv[1][0])[:40]:
    print('%5d x %12d el  %-10s %s' % (n, el, key[0], key[1]))
