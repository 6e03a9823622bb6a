"""HIP graphs for the fixed-shape, gradient-free segments of the iteration.

The teacher pass of the steady-state iteration (`TS_P2B_FCOS._teacher_pseudo`, fcos_p2b_teacher_student.py:227-244 in the reference:
`extract_feat` of the teacher + its dense head under no_grad) issues ~250 launches whose shapes depend on the image size only - not on
the ground truth.  With the LDS rings of round 5 the phase-2 iteration waits for Python (host issue ~20 ms against 19.8 ms of kernels),
so this segment is captured once per input signature and replayed: one host call instead of ~250.

Rules that make the capture safe ("streams and graphs"):
  * every launch of the C ABI goes to torch's current stream - the capture stream while capturing;
  * the segment reads parameters through pointers that are stable in the steady state: the flat parameter buffers, the weight planes of
    `functional._ConvWeightPlanes` (refreshed in place by ONE launch per parameter epoch, outside the segment), the persistent BatchNorm
    affine buffers (`nn_modules.refresh_bn_affines`), the census words.  The signature therefore contains the registration generation
    of the weight-plane cache, the identity of the affine buffers and the set of demoted fp16 groups: when any of them changes the
    segment is captured again;
  * inputs are copied into static buffers, outputs are static buffers - the caller consumes them on the replaying stream before the
    next replay (the teacher's stream is joined every iteration);
  * nothing in the segment may synchronise or upload; a capture that fails (torch raises) turns the graph off for the process with a
    warning and the segment runs eagerly - the results are the same launches either way (bit-identical, tests/test_teacher_graph.py).
"""
import os
import warnings

import torch

from . import planes as PL

ENABLED = os.environ.get('PT_TEACHER_GRAPH', '1') == '1'


def _flatten(obj, flat, seen):
    """-> a spec that `_rebuild` turns back into the same structure around other tensors; tensors are appended to `flat` once each."""
    if torch.is_tensor(obj):
        i = seen.get(id(obj))
        if i is None:
            i = seen[id(obj)] = len(flat)
            flat.append(obj)
        return ('t', i)
    if isinstance(obj, PL.PlaneAct):
        return ('p', _flatten(obj.t, flat, seen), obj.B, obj.H, obj.W, obj.C, obj.relu, obj.gcarrier)
    if isinstance(obj, (list, tuple)):
        return ('l' if isinstance(obj, list) else 'u',) + tuple(_flatten(o, flat, seen) for o in obj)
    if obj is None or isinstance(obj, (int, float, bool, str)):
        return ('c', obj)
    raise TypeError(f'graphs: cannot pass {type(obj)} through a captured segment')


def _rebuild(spec, flat):
    k = spec[0]
    if k == 't':
        return flat[spec[1]]
    if k == 'p':
        return PL.PlaneAct(_rebuild(spec[1], flat), *spec[2:7], gcarrier=spec[7])
    if k == 'l':
        return [_rebuild(s, flat) for s in spec[1:]]
    if k == 'u':
        return tuple(_rebuild(s, flat) for s in spec[1:])
    return spec[1]


class GraphedNoGrad:
    """`fn(*args)` (tensors, PlaneActs, lists / tuples of them, constants; no gradient) as a HIP graph per input signature after
    `warmup` eager calls.  `state_key()` -> hashable: everything outside the arguments whose change invalidates captured pointers."""

    def __init__(self, fn, state_key, name='segment', warmup=2):
        self.fn, self.state_key, self.name, self.warmup = fn, state_key, name, warmup
        self.entries = {}          # signature -> [eager calls so far, graph, static inputs, output structure]
        self.disabled = not ENABLED
        self.replays = 0
        self._stream = None        # the capture stream

    def __call__(self, *args):
        if self.disabled:
            return self.fn(*args)
        flat, spec = [], None
        spec = _flatten(args, flat, {})
        if not flat or not all(t.is_cuda for t in flat) or torch.is_grad_enabled() and any(t.requires_grad for t in flat):
            return self.fn(*args)
        sig = (spec, tuple((tuple(t.shape), t.dtype, t.stride()) for t in flat), self.state_key())
        e = self.entries.get(sig)
        if e is None:
            if len(self.entries) >= 4:                 # signatures that keep changing (a new image size every batch): not worth graphs
                self.entries.clear()
            e = self.entries[sig] = [0, None, None, None]
        if e[1] is None:
            e[0] += 1
            if e[0] <= self.warmup:
                return self.fn(*args)
            try:
                static = [torch.empty_like(t) for t in flat]
                for s, t in zip(static, flat):
                    s.copy_(t)
                # (capture_begin / capture_end on a stream of our own, not `torch.cuda.graph`: that context synchronises the device,
                #  collects garbage and EMPTIES the allocator's cache on entry - tens of ms of re-allocation in the next iteration)
                g = torch.cuda.CUDAGraph()
                cur = torch.cuda.current_stream()
                if self._stream is None:
                    self._stream = torch.cuda.Stream()
                self._stream.wait_stream(cur)
                with torch.cuda.stream(self._stream):
                    # thread_local: calls of OTHER threads (RCCL's watchdog of a multi-rank run) do not invalidate this capture
                    g.capture_begin(capture_error_mode='thread_local')
                    try:
                        out = self.fn(*_rebuild(spec, static))
                    finally:
                        g.capture_end()
                cur.wait_stream(self._stream)
                e[1], e[2], e[3] = g, static, out
            except Exception as ex:                   # an op that synchronises / uploads inside the segment: run eagerly from now on
                self.disabled = True
                self.entries.clear()
                warnings.warn(f'HIP graph capture of {self.name} failed ({type(ex).__name__}: {ex}); the segment runs eagerly',
                              RuntimeWarning, stacklevel=2)
                torch.cuda.synchronize()
                return self.fn(*args)
        else:
            for s, t in zip(e[2], flat):
                if s.data_ptr() != t.data_ptr():
                    s.copy_(t)
        e[1].replay()
        self.replays += 1
        return e[3]
