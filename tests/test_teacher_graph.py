"""The teacher pass as a HIP graph (point_teacher_amd/graphs.py): the replayed segment issues the launches of the eager one, so its
outputs differ from an eager run no more than two eager runs differ from each other - across the capture (warm-up calls, capture, replays), across parameter updates (the weight planes and
BatchNorm affines the graph reads are refreshed in place) and for both input forms (images / the shared frozen stem)."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(o, out):
    if torch.is_tensor(o):
        out.append(o)
    elif isinstance(o, (list, tuple)):
        for x in o:
            _flat(x, out)
    return out


@pytest.mark.gpu
def test_teacher_graph_matches_the_eager_pass():
    import point_teacher_amd as pta
    from point_teacher_amd import functional as F, graphs
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    dev = torch.device('cuda:0')
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    cfg.model['burn_in_step'] = -1
    torch.manual_seed(5)
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    model.train()
    tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
    data = SyntheticTiles(n=4, size=256, mean_objects=12, seed=3, device=dev)
    assert graphs.ENABLED
    for it in range(10):          # the signature settles once every weight form is registered; then 2 eager calls, the capture, replays
        tr.step(data.batch(it, 2))
    g = model._teacher_graph
    assert g is not None and not g.disabled and g.replays >= 2, (g and g.replays)
    img = data.batch(0, 2)['img'].contiguous(memory_format=torch.channels_last)

    def both(*args):
        with torch.no_grad():
            F.refresh_conv_weight_planes()
            r0 = g.replays
            a = [t.clone() for t in _flat(g(*args), [])]
            replayed = g.replays > r0
            b = [t.clone() for t in _flat(model._teacher_trunk(*args), [])]
            c = _flat(model._teacher_trunk(*args), [])
        assert len(a) == len(b) and len(a) >= 3
        d_ge = max(float((x.double() - y.double()).abs().max()) for x, y in zip(a, b))
        d_ee = max(float((x.double() - y.double()).abs().max()) for x, y in zip(b, c))
        print(f'graph vs eager max |d| {d_ge:.3e}; eager vs eager {d_ee:.3e}')
        # the same launches: the only difference is the one two EAGER runs show (the vendor library's split-k convolutions of the
        # 8 / 4 / 1-channel output maps add their partial sums atomically); stale weights would show at the 1e-3 level
        for x, y in zip(a, b):
            assert x.shape == y.shape
        assert d_ge <= max(3 * d_ee, 2e-5), (d_ge, d_ee)
        return replayed

    # the form the iteration used (the shared stem when the teacher shares the student's frozen stem, else the images)
    if model._shared_stem():
        t_stem = model._student_inputs(img)[4]            # exactly what forward_train hands to the teacher
        args = (None, t_stem)
    else:
        args = (img, None)
    assert both(*args), 'the signature of the live iteration must hit the captured graph'
    tr.step(data.batch(7, 2))                             # parameters move (SGD + EMA): same addresses, new values
    assert both(*args)
    # the other input form: three eager calls, then a capture of its own
    other = (img, None) if args[0] is None else None
    if other is not None:
        for _ in range(5):
            both(*other)
        assert both(*other)
    print('teacher graph: replays', g.replays, 'signatures', len(g.entries))


def test_graph_structure_round_trip():
    """Flatten / rebuild of the argument structure (tensors shared between positions stay shared)."""
    from point_teacher_amd import graphs, planes as PL
    a, b = torch.zeros(3), torch.ones(2, 2)
    pa = PL.PlaneAct(b, 1, 2, 1, 2, True)
    flat = []
    spec = graphs._flatten((None, (pa, [a, b, a]), 3), flat, {})
    assert len(flat) == 2
    r = graphs._rebuild(spec, [t + 1 for t in flat])
    assert r[0] is None and r[2] == 3 and isinstance(r[1][0], PL.PlaneAct) and r[1][0].relu and r[1][0].t is r[1][1][1]
    assert r[1][1][0] is r[1][1][2] and float(r[1][1][0].sum()) == 3.0
