"""Graph-net recurrences as persistent-kernel PROGRAMS (csrc/graph_prog.hip, include/ag2v.h "graph-net programs").

The time-step loops of ``Acts2LayoutModel.forward`` (reference models/graph_models/model.py:126-169) and
``MultiscaleActionDiscriminator.get_obj_vecs`` (discriminator.py:273-313) -- obj_vecs_net, 2-3 ``GraphTripleConv`` layers
(graph.py:41-107) and box_net per step, 15 steps -- are written here as ONE operator program per pass (forward; backward = the
input-gradient chain in reverse) that a single launch of co-resident workgroups interprets.  The weight gradients do not depend on
each other: after the backward program they are one GEMM per layer over the stacked time steps.

Everything a program touches lives in one arena allocated per pass (no tensor is reused, so the forward values are the backward's
ReLU masks); operands are row-major views described by (base, offset, rows, cols, leading dimension).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, ops

GEMM, CAT, POOL, POOL_BWD, GATHER_BWD, TRANSPOSE, ADD3 = 1, 2, 3, 4, 5, 6, 7

_OP_DTYPE = np.dtype([('type', '<i4'), ('M', '<i4'), ('N', '<i4'), ('K', '<i4'),
                      ('a', '<u8'), ('b', '<u8'), ('d', '<u8'), ('bias', '<u8'), ('res', '<u8'), ('mask', '<u8'), ('c', '<u8'),
                      ('i0', '<u8'), ('i1', '<u8'), ('i2', '<u8'), ('valid', '<u8'),
                      ('lda', '<i8'), ('ldb', '<i8'), ('ldd', '<i8'), ('ldr', '<i8'), ('ldm', '<i8'), ('ldc', '<i8'),
                      ('w0', '<i4'), ('w1', '<i4'), ('w2', '<i4'), ('relu', '<i4'), ('acc', '<i4'), ('O', '<i4'), ('T', '<i4'),
                      ('pad', '<i4', (3,))])


# Workgroups per program launch (None: the library's default, 128).  engine.TrainStep sets 64 when it co-schedules the three forward
# programs of a step on three streams: a program's workgroups must all be resident to pass its grid barriers (one 1024-thread
# workgroup with 135 KB of LDS per CU), so three concurrent programs must fit the chip together -- 3 x 64 of 256 CUs, with room for
# RCCL's kernels -- or two of them could hold each other's CUs (bounded by the barrier timeout, but a lost step).  The step time does
# not depend on the grid between 48 and 128 (tools/ab_graph_grid.sh: the operators have <= 48 tiles; the barrier is the cost).
GRID = [None]


def enabled():
    return os.environ.get('AG2V_NO_GRAPH_PROGRAM') is None


_MAX_EDGES = []


def max_edges():
    """Edges per sample the POOL / GATHER_BWD operators take (one 64-bit incidence mask per endpoint): the library's limit."""
    if not _MAX_EDGES:
        _MAX_EDGES.append(int(_lib.query('ag2v_graph_program_max_edges')))
    return _MAX_EDGES[0]


# ---- the timeout flag of every run is read back: a launch whose workgroups were not all co-resident gives up at a grid
# barrier after ~0.3 s, sets sync[1] and carries on with incomplete operands.  Each run copies its flag asynchronously into one slot of
# a pinned ring; check_timeouts() -- called by the training step next to its one host read (engine.TrainStep.gd_step: the NaN guard) and
# by anything else that is about to trust results -- waits for the copies made so far (long complete at that point) and raises.
_RING_SLOTS = 64
_ring = {'host': None, 'pending': [], 'next': 0}


def _watch_flag(prog):
    if _ring['host'] is None:
        _ring['host'] = torch.zeros(_RING_SLOTS, dtype=torch.int32).pin_memory()
    if len(_ring['pending']) >= _RING_SLOTS:
        check_timeouts()                          # the ring is full (nobody checked for 64 runs): drain it now, with a host wait
    slot = _ring['next'] % _RING_SLOTS
    _ring['next'] += 1
    _ring['host'][slot:slot + 1].copy_(prog.sync[1:2], non_blocking=True)
    _ring['pending'].append((slot, torch.cuda.current_stream().record_event(), prog.n_ops))


def check_timeouts():
    """Raises RuntimeError if a grid barrier of any graph-net program run since the last check gave up (that run's object vectors,
    boxes and gradients are invalid).  Waits only for the 4-byte flag copies of those runs."""
    pend, _ring['pending'] = _ring['pending'], []
    bad = []
    for slot, ev, n_ops in pend:
        ev.synchronize()
        if int(_ring['host'][slot]) != 0:
            bad.append(n_ops)
    if bad:
        raise RuntimeError("graph-net program: a grid barrier timed out in %d run(s) (programs of %s operators): the launch's workgroups "
                           "were not co-resident (another process or stream held the CUs); the results of those runs are invalid" % (len(bad), bad))


class Ref(object):
    """A row-major matrix view [rows, cols] with leading dimension ld at element offset `off` of `base` (a torch tensor, or None =
    the program's arena)."""
    __slots__ = ('base', 'off', 'rows', 'cols', 'ld')

    def __init__(self, base, off, rows, cols, ld):
        self.base, self.off, self.rows, self.cols, self.ld = base, int(off), int(rows), int(cols), int(ld)

    def cols_from(self, c0, n):
        return Ref(self.base, self.off + c0, self.rows, n, self.ld)


class Stack(object):
    """[steps, rows, cols] in the arena, one slot per time step (the weight-gradient GEMMs read the whole stack).  Slots start on
    128-byte boundaries -- a slot is written at its own step while its neighbours are read at others, and tensors must not share a
    cache line (csrc/graph_prog.hip) --, i.e. the stack is contiguous when rows * cols is a multiple of 32 (every c2 / c3 / c5 shape)."""

    def __init__(self, off, steps, rows, cols, ld=None):
        self.off, self.steps, self.rows, self.cols = off, steps, rows, cols
        self.ld = cols if ld is None else ld
        self.slot = (rows * self.ld + 31) & ~31

    def at(self, t):
        return Ref(None, self.off + t * self.slot, self.rows, self.cols, self.ld)


class Program(object):
    def __init__(self, device):
        self.device = device
        self.recs = []
        self.size = 0
        self.keep = []
        self.arena = None

    # ---- memory
    def new(self, rows, cols):
        off = self.size
        self.size += (rows * cols + 31) & ~31          # 128-byte blocks: two tensors never share a cache line (csrc/graph_prog.hip)
        return Ref(None, off, rows, cols, cols)

    def stack(self, steps, rows, cols, gemm_out=False):
        """gemm_out: the slots are written by GEMM tiles (32 columns each, by different workgroups): rows wider than one tile get a
        leading dimension that is a multiple of 32, so that no 128-byte line is shared by two tiles (csrc/graph_prog.hip: tensors AND
        tiles of different workgroups must not share a cache line -- a 516-column output once left stale halves in other XCDs' L2)."""
        off = self.size
        ld = ((cols + 31) & ~31) if (gemm_out and cols > 32 and cols % 32) else None
        st = Stack(off, steps, rows, cols, ld)
        self.size += steps * st.slot
        return st

    def ext(self, t, rows=None, cols=None):
        """An external contiguous fp32 tensor viewed as [rows, cols] (default: all leading dimensions folded into rows)."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("graph program: operands must be contiguous fp32 device tensors")
        self.keep.append(t)
        cols = t.shape[-1] if cols is None else cols
        rows = t.numel() // cols if rows is None else rows
        return Ref(t, 0, rows, cols, cols)

    def ext_at(self, t, index, rows, cols):
        """Slot `index` of a [steps, rows, cols] external stack."""
        self.keep.append(t)
        return Ref(t, index * rows * cols, rows, cols, cols)

    def _p(self, r):
        if r is None:
            return 0
        if isinstance(r, Ref):
            base = self._arena_ptr if r.base is None else r.base.data_ptr()
            return base + 4 * r.off
        self.keep.append(r)
        return r.data_ptr()          # raw tensor (index / validity / bias arrays)

    # ---- operators (recorded; pointers are resolved in finalize)
    def _rec(self, **kw):
        self.recs.append(kw)

    @staticmethod
    def _need(cond, msg):
        # (explicit raises, not asserts: under python -O a violated constraint would come back as stale cache lines or misaligned
        # 16-byte accesses inside the kernel)
        if not cond:
            raise RuntimeError("graph program: " + msg)

    def _aligned(self, *refs):
        for r in refs:
            if isinstance(r, Ref):
                self._need(r.cols % 4 == 0 and r.ld % 4 == 0 and r.off % 4 == 0,
                           "widths, leading dimensions and offsets must be multiples of 4 floats (got cols %d, ld %d, off %d)" % (r.cols, r.ld, r.off))

    def gemm(self, a, b, out, bias=None, res=None, mask=None, relu=False, acc=False):
        self._need(a.cols == b.cols and out.rows == a.rows and out.cols == b.rows,
                   "GEMM shapes %r" % ((a.rows, a.cols, b.rows, b.cols, out.rows, out.cols),))
        self._need(out.cols <= 32 or out.ld % 32 == 0, "GEMM outputs wider than a tile need a leading dimension that is a multiple of 32")
        self._aligned(a, b, out, res, mask)
        self._rec(type=GEMM, M=a.rows, N=b.rows, K=a.cols, a=a, b=b, bias=bias, res=res, mask=mask, c=out, relu=int(relu), acc=int(acc))

    def cat(self, out, segs):
        """segs: up to three (Ref, row-index int32 tensor or None)."""
        kw = dict(type=CAT, M=out.rows, c=out)
        names = (('a', 'i0', 'w0'), ('b', 'i1', 'w1'), ('d', 'i2', 'w2'))
        self._need(sum(s[0].cols for s in segs) == out.cols, "CAT segment widths do not add up to the output width")
        self._aligned(out, *[s[0] for s in segs])
        for (ref, idx), (pn, inn, wn) in zip(segs, names):
            kw[pn], kw[inn], kw[wn] = ref, idx, ref.cols
        self._rec(**kw)

    def _edges_ok(self, T, *refs):
        self._need(T <= max_edges(), "%d edges per sample; POOL / GATHER_BWD take at most %d (csrc/graph_prog.hip)" % (T, max_edges()))
        self._aligned(*refs)

    def pool(self, newt, out, counts, i0, i1, valid, O, T, Hd, Dp):
        self._edges_ok(T, newt, out)
        self._need(Hd % 4 == 0 and Dp % 4 == 0, "POOL widths must be multiples of 4")
        self._rec(type=POOL, M=out.rows, a=newt, c=out, mask=counts, i0=i0, i1=i1, valid=valid, O=O, T=T, w0=Hd, w1=Dp)

    def pool_bwd(self, dpooled, dp_in, out, counts, i0, i1, valid, O, T, Hd, Dp):
        self._edges_ok(T, dpooled, dp_in, out)
        self._rec(type=POOL_BWD, M=out.rows, a=dpooled, b=dp_in, c=out, mask=counts, i0=i0, i1=i1, valid=valid, O=O, T=T, w0=Hd, w1=Dp)

    def gather_bwd(self, dtri, out, i0, i1, O, T, Din, Dp, res=None):
        self._edges_ok(T, dtri, out, res)
        self._need(Din % 4 == 0 and Dp % 4 == 0, "GATHER_BWD widths must be multiples of 4")
        self._rec(type=GATHER_BWD, M=out.rows, a=dtri, c=out, res=res, i0=i0, i1=i1, O=O, T=T, w0=Din, w1=Dp)

    def transpose(self, w, out):
        self._rec(type=TRANSPOSE, N=w.rows, K=w.cols, a=w, c=out)

    def add3(self, out, a, b=None, d=None):
        self._rec(type=ADD3, M=out.rows, N=out.cols, a=a, b=b, d=d, c=out)

    # ---- build + run
    def finalize(self):
        self.arena = torch.empty(max(self.size, 4), device=self.device, dtype=torch.float32)
        self._arena_ptr = self.arena.data_ptr()
        arr = np.zeros(len(self.recs), _OP_DTYPE)
        self._need(_OP_DTYPE.itemsize == _lib.query('ag2v_graph_op_bytes'), "operator record layout differs from the library's")
        lds = {'a': 'lda', 'b': 'ldb', 'd': 'ldd', 'res': 'ldr', 'mask': 'ldm', 'c': 'ldc'}
        for i, r in enumerate(self.recs):
            for k, v in r.items():
                if k in ('a', 'b', 'd', 'bias', 'res', 'mask', 'c', 'i0', 'i1', 'i2', 'valid'):
                    arr[i][k] = self._p(v)
                    if k in lds and isinstance(v, Ref):
                        arr[i][lds[k]] = v.ld
                else:
                    arr[i][k] = v
        self.n_ops = len(self.recs)
        self.ops_dev = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(self.device, non_blocking=False)
        self.sync = torch.zeros(4, device=self.device, dtype=torch.int32)
        self.recs = None
        return self

    def run(self):
        grid = int(os.environ.get('AG2V_GRAPH_GRID', 0)) or GRID[0] or _lib.query('ag2v_graph_program_grid')
        _lib.call('ag2v_graph_program_run', ctypes.c_void_p(self.ops_dev.data_ptr()), self.n_ops, ctypes.c_void_p(self.sync.data_ptr()),
                  grid, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _watch_flag(self)

    def timed_out(self):
        """Host read (synchronises): True when a grid barrier of the LAST run gave up -- the results of that run are invalid."""
        return bool(self.sync[1].item())

    def tensor(self, ref_or_stack):
        if isinstance(ref_or_stack, Stack):
            s = ref_or_stack
            return torch.as_strided(self.arena, (s.steps, s.rows, s.cols), (s.slot, s.ld, 1), s.off)
        r = ref_or_stack
        assert r.base is None
        return torch.as_strided(self.arena, (r.rows, r.cols), (r.ld, 1), r.off)


# ------------------------------------------------------------------------------------------------------------------------------------
# The recurrent net.  Parameters: first = (W1, W2): obj_vecs_net (no bias); per GraphTripleConv (n1a_w, n1a_b, n1b_w, n1b_b, n2a_w,
# n2a_b, n2b_w, n2b_b); box_net (b1_w, b1_b, b2_w, b2_b) or none.
# mode 'emb': step input = cat(emb, boxes[t-1]) and boxes[t] = boxes[t-1] + box_net(ov_t)  (Acts2LayoutModel)
# mode 'ov' : step input = cat(ov_{t-1}, boxes_given[t])                                    (the discriminator's get_obj_vecs)
#
# A PLAN holds the two programs of one (shape, parameter storage) configuration with every operand at a fixed address -- the inputs
# are copied into slots of the plan -- so the operator arrays are written and uploaded ONCE and a training step only copies a few
# small inputs and launches.  A plan is busy from its forward until its backward (or until the forward's output is dropped); a
# second forward in between takes another plan of the pool.
class _Plan(object):
    def __init__(self, mode, S, BO, BT, O, T, n_gconv, has_box, x0, pv_all, params):
        dev = x0.device
        self.dev, self.mode, self.S, self.BO, self.BT, self.O, self.T = dev, mode, S, BO, BT, O, T
        self.n_gconv, self.has_box = n_gconv, has_box
        self.busy, self.out_ref = False, None
        W1, W2 = params[0], params[1]
        gcs = [params[2 + 8 * l: 10 + 8 * l] for l in range(n_gconv)]
        box = params[2 + 8 * n_gconv: 6 + 8 * n_gconv] if has_box else None
        E, dp0 = x0.shape[1], pv_all.shape[2]
        Hd = gcs[0][2].shape[1]                 # hidden_dim (n1b_w: [2 Hd + Dp, Hd])
        Dp = gcs[0][2].shape[0] - 2 * Hd
        Dout = gcs[-1][6].shape[0]
        # input slots (fixed addresses)
        self.x0 = torch.empty((BO, E), device=dev)
        self.pv = torch.empty((S, BT, dp0), device=dev)
        self.boxes_in = torch.empty((BO, 4) if mode == 'emb' else (S, BO, 4), device=dev)
        self.i0 = torch.empty((S, BT), device=dev, dtype=torch.int32)
        self.i1 = torch.empty((S, BT), device=dev, dtype=torch.int32)
        self.valid = torch.empty((S, BT), device=dev, dtype=torch.uint8)
        self.d_ov = torch.zeros((S, BO, Dout), device=dev)
        self.d_boxes = torch.zeros((S, BO, 4), device=dev) if has_box else None
        i0s, i1s, vls = list(self.i0.unbind(0)), list(self.i1.unbind(0)), list(self.valid.unbind(0))
        P = Program(dev)
        # (every GEMM-written stack is allocated gemm_out: a width that is not a multiple of 32 gets a padded leading dimension instead
        # of tripping the tile / cache-line rule; the weight-gradient GEMMs read the stacks through their strides)
        f = {'in0': P.stack(S, BO, W1.shape[1]), 'h1': P.stack(S, BO, W1.shape[0], gemm_out=True), 'ov0': P.stack(S, BO, W2.shape[0], gemm_out=True)}
        Dins = []
        for l, g in enumerate(gcs):
            dpi = dp0 if l == 0 else Dp
            Din = (g[0].shape[1] - dpi) // 2
            Dins.append(Din)
            f['cur%d' % l] = P.stack(S, BT, 2 * Din + dpi)
            f['a1%d' % l] = P.stack(S, BT, g[0].shape[0], gemm_out=True)
            f['nt%d' % l] = P.stack(S, BT, g[2].shape[0], gemm_out=True)
            f['pool%d' % l] = P.stack(S, BO, Hd)
            f['cnt%d' % l] = P.stack(S, BO, 1)
            f['a2%d' % l] = P.stack(S, BO, g[4].shape[0], gemm_out=True)
            f['ov%d' % (l + 1)] = P.stack(S, BO, g[6].shape[0], gemm_out=True)
        if has_box:
            f['bh'] = P.stack(S, BO, box[0].shape[0], gemm_out=True)
            f['boxes'] = P.stack(S, BO, 4)
        x0r = P.ext(self.x0, BO, E)
        for s in range(S):
            i0, i1, vl = i0s[s], i1s[s], vls[s]
            if mode == 'emb':
                bprev = P.ext(self.boxes_in, BO, 4) if s == 0 else f['boxes'].at(s - 1)
                P.cat(f['in0'].at(s), [(x0r, None), (bprev, None)])
            else:
                ovprev = x0r if s == 0 else f['ov%d' % n_gconv].at(s - 1)
                P.cat(f['in0'].at(s), [(ovprev, None), (P.ext_at(self.boxes_in, s, BO, 4), None)])
            P.gemm(f['in0'].at(s), P.ext(W1), f['h1'].at(s), relu=True)
            P.gemm(f['h1'].at(s), P.ext(W2), f['ov0'].at(s), relu=True)
            ov = f['ov0'].at(s)
            pv = P.ext_at(self.pv, s, BT, dp0)
            for l, g in enumerate(gcs):
                n1a_w, n1a_b, n1b_w, n1b_b, n2a_w, n2a_b, n2b_w, n2b_b = g
                P.cat(f['cur%d' % l].at(s), [(ov, i0), (pv, None), (ov, i1)])
                P.gemm(f['cur%d' % l].at(s), P.ext(n1a_w), f['a1%d' % l].at(s), bias=n1a_b, relu=True)
                P.gemm(f['a1%d' % l].at(s), P.ext(n1b_w), f['nt%d' % l].at(s), bias=n1b_b, relu=True)
                P.pool(f['nt%d' % l].at(s), f['pool%d' % l].at(s), f['cnt%d' % l].at(s), i0, i1, vl, O, T, Hd, Dp)
                P.gemm(f['pool%d' % l].at(s), P.ext(n2a_w), f['a2%d' % l].at(s), bias=n2a_b, relu=True)
                P.gemm(f['a2%d' % l].at(s), P.ext(n2b_w), f['ov%d' % (l + 1)].at(s), bias=n2b_b, relu=True)
                ov = f['ov%d' % (l + 1)].at(s)
                pv = f['nt%d' % l].at(s).cols_from(Hd, Dp)
            if has_box:
                bprev = P.ext(self.boxes_in, BO, 4) if s == 0 else f['boxes'].at(s - 1)
                P.gemm(ov, P.ext(box[0]), f['bh'].at(s), bias=box[1], relu=True)
                P.gemm(f['bh'].at(s), P.ext(box[2]), f['boxes'].at(s), bias=box[3], res=bprev)
        self.fwd, self.f = P.finalize(), f
        self.dims = (E, Hd, Dp, Dout, Dins, dp0)
        self.bwd = self.g = None
        self._params = params

    def build_backward(self):
        mode, S, BO, BT, O, T, n_gconv, has_box = self.mode, self.S, self.BO, self.BT, self.O, self.T, self.n_gconv, self.has_box
        params = self._params
        W1, W2 = params[0], params[1]
        gcs = [params[2 + 8 * l: 10 + 8 * l] for l in range(n_gconv)]
        box = params[2 + 8 * n_gconv: 6 + 8 * n_gconv] if has_box else None
        E, Hd, Dp, Dout, Dins, dp0 = self.dims
        F, f = self.fwd, self.f
        i0s, i1s, vls = list(self.i0.unbind(0)), list(self.i1.unbind(0)), list(self.valid.unbind(0))

        def fr(name, s):        # a forward value (lives in the forward program's arena)
            r = f[name].at(s)
            return Ref(F.arena, r.off, r.rows, r.cols, r.ld)
        P = Program(self.dev)
        P.keep.append(F.arena)

        def tr(w):              # transposed weights: the input-gradient GEMMs reduce over the output index
            out = P.new(w.shape[1], w.shape[0])
            P.transpose(P.ext(w), out)
            return out
        W1t, W2t = tr(W1), tr(W2)
        gct = [(tr(g[0]), tr(g[2]), tr(g[4]), tr(g[6])) for g in gcs]
        boxt = (tr(box[0]), tr(box[2])) if has_box else None
        g = {'din0': P.stack(S, BO, W1.shape[1], gemm_out=True), 'dh1': P.stack(S, BO, W1.shape[0], gemm_out=True),
             'dov0': P.stack(S, BO, W2.shape[0])}
        for l in range(n_gconv):
            dpi = dp0 if l == 0 else Dp
            g['dcur%d' % l] = P.stack(S, BT, 2 * Dins[l] + dpi, gemm_out=True)
            g['da1%d' % l] = P.stack(S, BT, gcs[l][0].shape[0], gemm_out=True)
            g['dnt%d' % l] = P.stack(S, BT, gcs[l][2].shape[0])
            g['dpool%d' % l] = P.stack(S, BO, Hd, gemm_out=True)
            g['da2%d' % l] = P.stack(S, BO, gcs[l][4].shape[0], gemm_out=True)
            g['dov%d' % (l + 1)] = P.stack(S, BO, gcs[l][6].shape[0], gemm_out=True)
        if has_box:
            g['dbh'] = P.stack(S, BO, box[0].shape[0], gemm_out=True)
            g['gbox'] = P.stack(S, BO, 4)
        for s in range(S - 1, -1, -1):
            i0, i1, vl = i0s[s], i1s[s], vls[s]
            top = g['dov%d' % n_gconv].at(s)
            dov_ext = P.ext_at(self.d_ov, s, BO, Dout)
            if has_box:
                # gradient of boxes[s]: the output's, plus (s < S - 1) the residual path and the next step's input columns
                if s == S - 1:
                    P.add3(g['gbox'].at(s), P.ext_at(self.d_boxes, s, BO, 4))
                else:
                    P.add3(g['gbox'].at(s), P.ext_at(self.d_boxes, s, BO, 4), g['gbox'].at(s + 1), g['din0'].at(s + 1).cols_from(E, 4))
                P.gemm(g['gbox'].at(s), boxt[1], g['dbh'].at(s))
                P.gemm(g['dbh'].at(s), boxt[0], top, mask=fr('bh', s), res=dov_ext)
            elif mode == 'ov' and s < S - 1:
                P.add3(top, dov_ext, g['din0'].at(s + 1).cols_from(0, Dout))
            else:
                P.add3(top, dov_ext)
            for l in range(n_gconv - 1, -1, -1):
                n1a_t, n1b_t, n2a_t, n2b_t = gct[l]
                P.gemm(g['dov%d' % (l + 1)].at(s), n2b_t, g['da2%d' % l].at(s), mask=fr('ov%d' % (l + 1), s))
                P.gemm(g['da2%d' % l].at(s), n2a_t, g['dpool%d' % l].at(s), mask=fr('a2%d' % l, s))
                dp_in = None
                if l < n_gconv - 1:
                    dp_in = g['dcur%d' % (l + 1)].at(s).cols_from(Dins[l + 1], Dp)
                P.pool_bwd(g['dpool%d' % l].at(s), dp_in, g['dnt%d' % l].at(s), fr('cnt%d' % l, s), i0, i1, vl, O, T, Hd, Dp)
                P.gemm(g['dnt%d' % l].at(s), n1b_t, g['da1%d' % l].at(s), mask=fr('nt%d' % l, s))
                P.gemm(g['da1%d' % l].at(s), n1a_t, g['dcur%d' % l].at(s), mask=fr('a1%d' % l, s))
                P.gather_bwd(g['dcur%d' % l].at(s), g['dov%d' % l].at(s), i0, i1, O, T, Dins[l], dp0 if l == 0 else Dp)
            P.gemm(g['dov0'].at(s), W2t, g['dh1'].at(s), mask=fr('ov0', s))
            P.gemm(g['dh1'].at(s), W1t, g['din0'].at(s), mask=fr('h1', s))
        self.bwd, self.g = P.finalize(), g


import collections

_PLANS = collections.OrderedDict()       # key -> pool of plans; least recently used keys are dropped beyond _MAX_PLAN_KEYS
_MAX_PLAN_KEYS = 12                      # a training run has 3 nets x (clip step, graph step) = 6 keys once T is bucketed (run_recurrence)


def _get_plan(key, make):
    pool = _PLANS.get(key)
    if pool is None:
        pool = _PLANS[key] = []
        while len(_PLANS) > _MAX_PLAN_KEYS:
            old_key = next(iter(_PLANS))
            if any(pl.busy and pl.out_ref is not None and pl.out_ref() is not None for pl in _PLANS[old_key]):
                break                        # its backward is still to come: keep it (checked again at the next insertion)
            del _PLANS[old_key]              # arenas, operator arrays and parameter references of that shape are released
    else:
        _PLANS.move_to_end(key)
    for pl in pool:
        if pl.busy and pl.out_ref is not None and pl.out_ref() is None:
            pl.busy = False                  # the forward's autograd node is gone: nobody can back-propagate into this plan any more
        if not pl.busy:
            return pl
    pl = make()
    if len(pool) < 4:
        pool.append(pl)
    return pl


class _Recurrence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, x0, pv_all, boxes_in, *params):
        """x0: emb [BO, E] (mode 'emb') or the initial object vectors [BO, D] (mode 'ov'); pv_all [S, BT, Dp] frame-major predicate /
        action vectors of the S recurrence steps; boxes_in: [BO, 4] initial boxes (mode 'emb') or [S, BO, 4] given boxes (mode 'ov').
        Returns (ov_stack [S, BO, Dout], boxes_stack [S, BO, 4] (mode 'emb') or an empty tensor)."""
        mode, S, BO, BT, O, T, i0, i1, valid, n_gconv, has_box = cfg
        params = [p.contiguous() for p in params]
        key = (mode, S, BO, BT, O, T, n_gconv, has_box, tuple(x0.shape), tuple(pv_all.shape), str(x0.device),
               tuple(p.data_ptr() for p in params))
        pl = _get_plan(key, lambda: _Plan(mode, S, BO, BT, O, T, n_gconv, has_box, x0, pv_all, params))
        pl.x0.copy_(x0)
        pl.pv.copy_(pv_all)
        pl.boxes_in.copy_(boxes_in)
        pl.i0.copy_(i0)
        pl.i1.copy_(i1)
        pl.valid.copy_(valid)
        pl.fwd.run()
        needs = any(ctx.needs_input_grad)
        pl.busy = bool(needs)
        ctx.pl = pl
        ctx.wkeys = [(ops._wsink_register(p) if ctx.needs_input_grad[4 + i] else None) for i, p in enumerate(params)]
        ctx.p_alive = params
        # the outputs are COPIES (S * BO * Dout floats): the arena belongs to the plan, whose next forward overwrites it -- a caller that
        # keeps an output (logging, metrics, retain_graph) must not see it change; the arena itself stays the source of the backward's masks
        ov_stack = pl.fwd.tensor(pl.f['ov%d' % n_gconv]).clone()
        boxes_stack = pl.fwd.tensor(pl.f['boxes']).clone() if has_box else torch.empty(0, device=x0.device)
        if needs:
            import weakref
            pl.out_ref = weakref.ref(ctx)    # the autograd node lives exactly as long as something can still back-propagate into it
        return ov_stack, boxes_stack

    @staticmethod
    def backward(ctx, d_ov, d_boxes):
        pl = ctx.pl
        if pl is None:
            raise RuntimeError("graph program: this recurrence has already been back-propagated (its plan -- the forward values the "
                               "backward masks with -- was released); run the forward again instead of a second backward / retain_graph")
        mode, S, BO, BT, n_gconv, has_box = pl.mode, pl.S, pl.BO, pl.BT, pl.n_gconv, pl.has_box
        E, Hd, Dp, Dout, Dins, dp0 = pl.dims
        dev = pl.dev
        if pl.bwd is None:
            pl.build_backward()
        if d_ov is None:
            pl.d_ov.zero_()
        else:
            pl.d_ov.copy_(d_ov)
        if has_box:
            if d_boxes is None:
                pl.d_boxes.zero_()
            else:
                pl.d_boxes.copy_(d_boxes)
        pl.bwd.run()
        P, F, f, g = pl.bwd, pl.fwd, pl.f, pl.g
        T_ = lambda name: P.tensor(g[name])
        FT = lambda name: F.tensor(f[name])
        st = ops._stream()
        params = ctx.p_alive
        grads = [None] * len(params)

        def lin_grads(wi, bi, A, dY, Y):
            """dW (+)= dZ^T A, db (+)= column sums of dZ, dZ = dY masked by the forward output Y (None: no ReLU) -- over all steps."""
            M = dY.shape[0] * dY.shape[1]
            N, K = dY.shape[2], A.shape[2]
            dZ = dY.reshape(M, N)
            if Y is not None:
                z = torch.empty_like(dZ)
                _lib.call('ag2v_relu_bwd', ops._ptr(dZ), ops._ptr(Y.reshape(M, N)), ops._ptr(z), dZ.numel(), st)
                dZ = z
            if ctx.needs_input_grad[4 + wi]:
                buf, acc, dw = ops._wsink_take(ctx.wkeys[wi], params[wi])
                _lib.call('ag2v_gemm_f32', ops._ptr(dZ), ops._ptr(A.reshape(M, K)), None, ops._ptr(buf), N, K, M, 1, N, K, 1, K, 0, acc, st)
                grads[wi] = dw
            if bi is not None and ctx.needs_input_grad[4 + bi]:
                bbuf, bacc, db = ops._wsink_take(ctx.wkeys[bi], params[bi])
                # (column sums over ~1 300 stacked rows: a tree reduction -- the graph kernels' one-thread-per-column col_sum walks the
                # rows serially, 165 us per layer here)
                if bacc:
                    bbuf.add_(dZ.sum(dim=0))
                else:
                    torch.sum(dZ, dim=0, out=bbuf)
                grads[bi] = db
        lin_grads(0, None, FT('in0'), T_('dh1'), FT('h1'))
        lin_grads(1, None, FT('h1'), T_('dov0'), FT('ov0'))
        for l in range(n_gconv):
            b0 = 2 + 8 * l
            lin_grads(b0, b0 + 1, FT('cur%d' % l), T_('da1%d' % l), FT('a1%d' % l))
            lin_grads(b0 + 2, b0 + 3, FT('a1%d' % l), T_('dnt%d' % l), FT('nt%d' % l))
            lin_grads(b0 + 4, b0 + 5, FT('pool%d' % l), T_('da2%d' % l), FT('a2%d' % l))
            lin_grads(b0 + 6, b0 + 7, FT('a2%d' % l), T_('dov%d' % (l + 1)), FT('ov%d' % (l + 1)))
        if has_box:
            b0 = 2 + 8 * n_gconv
            lin_grads(b0, b0 + 1, FT('ov%d' % n_gconv), T_('dbh'), FT('bh'))
            lin_grads(b0 + 2, b0 + 3, FT('bh'), T_('gbox'), None)
        din0 = T_('din0')
        dx0 = dpv = None
        if ctx.needs_input_grad[1]:
            dx0 = din0[:, :, :E].sum(dim=0) if mode == 'emb' else din0[0, :, :Dout].clone()
        if ctx.needs_input_grad[2]:
            dpv = T_('dcur0')[:, :, Dins[0]:Dins[0] + dp0].contiguous()
        pl.busy = False
        ctx.pl = None
        return (None, dx0, dpv, None) + tuple(grads)


def _row_indices(edges_all, O):
    """edges_all int64 [S, B, T, 2] (object index inside the sample) -> int32 row indices b * O + index, [S, B * T] each."""
    S, B, T, _ = edges_all.shape
    base = (torch.arange(B, device=edges_all.device, dtype=torch.int64) * O).view(1, B, 1)
    i0 = (edges_all[..., 0] + base).reshape(S, B * T).to(torch.int32).contiguous()
    i1 = (edges_all[..., 1] + base).reshape(S, B * T).to(torch.int32).contiguous()
    return i0, i1


def run_recurrence(mode, x0, pv_steps, edges_steps, ind_steps, boxes_in, first, gconvs, box):
    """x0 [B, O, E | D]; pv_steps [S, B, T, Dp]; edges_steps int64 [S, B, T, 2]; ind_steps bool [S, B, T]; boxes_in [B, O, 4]
    (mode 'emb': the first frame's boxes) or [S, B, O, 4] (mode 'ov'); first = obj_vecs_net's two Linear modules, gconvs = list of
    GraphTripleConv, box = box_net's two Linear modules or None.  Returns (ov [S, B, O, Dout], boxes [S, B, O, 4] or None)."""
    S, B, T = pv_steps.shape[0], pv_steps.shape[1], pv_steps.shape[2]
    O = x0.shape[1]
    if T > max_edges():
        raise RuntimeError("graph program: %d edges per sample exceed the kernel's %d; callers check supported(args, T) first" % (T, max_edges()))
    Tb = min(max_edges(), (T + 7) & ~7)
    if Tb != T:
        # T is the per-batch padded maximum of triplets + actions (collate_fn): bucketed to multiples of 8 with edges that are padding in
        # the kernel's sense too -- index 0, invalid (POOL skips them; their rows receive zero gradient, so GATHER_BWD adds zeros) -- one
        # plan per net serves every batch of a bucket instead of one plan (programs, arenas) per distinct T
        pad = Tb - T
        pv_steps = torch.nn.functional.pad(pv_steps, (0, 0, 0, pad))
        edges_steps = torch.nn.functional.pad(edges_steps, (0, 0, 0, pad))
        ind_steps = torch.nn.functional.pad(ind_steps, (0, pad))
        T = Tb
    i0, i1 = _row_indices(edges_steps, O)
    valid = ind_steps.reshape(S, B * T).to(torch.uint8).contiguous()
    params = [first[0].weight, first[1].weight]
    for gc in gconvs:
        lins = [m for m in gc.net1.children() if hasattr(m, 'weight')] + [m for m in gc.net2.children() if hasattr(m, 'weight')]
        for m in lins:
            params += [m.weight, m.bias]
    if box is not None:
        params += [box[0].weight, box[0].bias, box[1].weight, box[1].bias]
    cfg = (mode, S, B * O, B * T, O, T, i0, i1, valid, len(gconvs), box is not None)
    bi = boxes_in.reshape(B * O, 4) if mode == 'emb' else boxes_in.reshape(S, B * O, 4)
    ov, boxes = _Recurrence.apply(cfg, x0.reshape(B * O, -1), pv_steps.reshape(S, B * T, -1), bi, *params)
    ov = ov.reshape(S, B, O, -1)
    return ov, (boxes.reshape(S, B, O, 4) if box is not None else None)


def supported(module_or_opt, T=None):
    """True when the recurrence can run as a program: no MLP normalisation, average pooling, every width a multiple of 4 floats (the
    kernels' 16-byte accesses; GEMM outputs get padded leading dimensions where 32-column tiles need them) and at most max_edges() edges
    per sample (T = the padded triplets + actions of this batch, when the caller knows it).  Otherwise callers take the per-operator path."""
    args = module_or_opt if isinstance(module_or_opt, dict) else vars(module_or_opt)
    if not (enabled() and args.get('mlp_normalization', 'none') in ('none', None) and args.get('gconv_pooling', 'avg') == 'avg'):
        return False
    for k in ('embedding_dim', 'gconv_dim', 'gconv_hidden_dim'):
        if int(args.get(k, 128)) % 4:
            return False
    return T is None or int(T) <= max_edges()
