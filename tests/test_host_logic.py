"""Host-side logic of the hot path that needs no GPU: branch selection of the dense head (burn-in step 1 evaluates only the tower
whose outputs are read), the split ResNet forward the shared frozen stem relies on, the executed-FLOP accounting of bench.py
and the host path of the index-upload helper."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _head():
    import point_teacher_amd as pta
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    hc = dict(cfg.model['_model_']['bbox_head'])
    hc.update(train_cfg=cfg.model['train_cfg'], test_cfg=cfg.model['test_cfg'])
    torch.manual_seed(0)
    return pta.build_from_cfg(hc, pta.HEADS)


def test_head_branches_are_the_same_numbers():
    """`branches='reg'` / `'cls'` skip a tower and return None for its outputs; what they do return equals the full forward."""
    head = _head().eval()
    x = [torch.randn(2, 256, 12, 12)]
    with torch.no_grad():
        cls, reg, ctr, pts = head(x)
        cls_r, reg_r, ctr_r, pts_r = head(x, branches='reg')
        cls_c, reg_c, ctr_c, pts_c = head(x, branches='cls')
    assert cls_r[0] is None and reg_c[0] is None and ctr_c[0] is None
    assert torch.equal(reg_r[0], reg[0]) and torch.equal(ctr_r[0], ctr[0]) and torch.equal(cls_c[0], cls[0])
    assert torch.equal(pts_r[0], pts[0]) and torch.equal(pts_c[0], pts[0])
    fc, fr, ft = head._flat(cls_r, reg_r, ctr_r)
    assert fc is None and fr.shape == (2, 144, 4) and ft.shape == (2, 144)
    fc, fr, ft = head._flat(cls_c, reg_c, ctr_c)
    assert fr is None and ft is None and fc.shape == (2, 144, head.cls_out_channels)
    # gradients: the skipped tower receives none, the evaluated one the same as in the full pass
    head.train()
    head.zero_grad()
    head(x, branches='reg')[1][0].sum().backward()
    assert head.cls_convs[0].conv.weight.grad is None and head.conv_cls.weight.grad is None
    g_reg = head.reg_convs[0].conv.weight.grad.clone()
    head.zero_grad()
    head(x)[1][0].sum().backward()
    torch.testing.assert_close(head.reg_convs[0].conv.weight.grad, g_reg)


def test_resnet_split_forward():
    """forward(x) == forward(None, stem=forward_stem(x)); the stem is exactly the frozen part (no parameter in it trains)."""
    from point_teacher_amd.nn_modules import ResNet
    torch.manual_seed(1)
    net = ResNet(50, frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe').train()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        full = net(x)
        stem = net.forward_stem(x)
        split = net(None, stem=stem)
        rows = net(None, stem=(stem[0][:1], [o[:1] for o in stem[1]]))          # the teacher continues from the first rows
    assert len(full) == len(split) == 4
    for a, b, c in zip(full, split, rows):
        assert torch.equal(a, b)
        torch.testing.assert_close(a[:1], c, rtol=1e-3, atol=1e-4 * float(a.abs().max()))   # another batch size: another CPU conv kernel
    stem_params = [p for m in (net.conv1, net.bn1, net.layer1) for p in m.parameters()]
    assert stem_params and not any(p.requires_grad for p in stem_params)
    assert any(p.requires_grad for p in net.layer2.parameters())
    # nothing frozen: the "stem" is conv1 / bn1 / maxpool only and callers must not share it (TS_P2B_FCOS._shared_stem checks)
    free = ResNet(50, frozen_stages=-1, norm_eval=True)
    assert free.frozen_stages == -1 and any(p.requires_grad for p in free.conv1.parameters())


def test_iteration_flops_accounting():
    import point_teacher_amd as pta
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py')).to_dict()['model']
    f1, f1r = bench.iteration_flops('step1', cfg, 2, 800, 300), bench.iteration_flops('step1', cfg, 2, 800, 300, executed=False)
    f2, f2r = bench.iteration_flops('step2', cfg, 2, 800, 300), bench.iteration_flops('step2', cfg, 2, 800, 300, executed=False)
    trunk, stem, head = 60.2e9, 10.2e9, 47.5e9
    # the reference additionally runs the teacher's stem (both phases) and the forward of one unread tower per step-1 head pass
    assert f2r - f2 == pytest.approx(2.0 * 2 * stem, rel=1e-9)
    assert f1r - f1 == pytest.approx(2.0 * 2 * (stem + 2 * 0.5 * head), rel=1e-9)
    assert 3.8e12 < f1 < 4.0e12 and 3.0e12 < f2 < 3.2e12
    # size and batch scale the convolution part only as area x batch
    assert bench.iteration_flops('step2', cfg, 4, 800, 300) == pytest.approx(2 * f2, rel=1e-9)


def test_upload_helper_host_path():
    from point_teacher_amd import functional as Fn
    a = Fn.upload_i32([3, 1, 4], 'cpu')
    b = Fn.upload_f32(np.array([1.5, 2.5]), 'cpu')
    c = Fn.upload_i32([9, 9, 9], 'cpu')
    assert a.tolist() == [3, 1, 4] and a.dtype == torch.int32 and b.tolist() == [1.5, 2.5] and c.tolist() == [9, 9, 9]
    off, tot = Fn.make_offsets([2, 0, 5], 'cpu')
    assert off.tolist() == [0, 2, 2, 7] and tot == 7
    assert Fn._ring.seg_words * Fn._ring.SEG == Fn._ring.words


def test_pinned_ring_cursor_exact_fills():
    """an upload that ends exactly on a segment end - or on the end of the ring - used to park the cursor
    there: no event for the segment left, no wait for the segment entered, and at the ring's end an empty slice (ValueError)
    that broke every later upload.  The cursor logic is driven here without a device: events are a log."""
    from point_teacher_amd.functional import _PinnedRing

    class Ring(_PinnedRing):
        SEG = 4

        def __init__(self):
            super().__init__(words=64)                      # 4 segments of 16 words
            self.log = []
            self.buf = torch.zeros(self.words, dtype=torch.int32)

        def _record(self, seg, stream):
            self.log.append(('record', seg))

        def _await(self, seg):
            self.log.append(('await', seg))

        def put(self, n, dtype=torch.int32):
            arr = np.arange(n, dtype=np.int32 if dtype == torch.int32 else np.float32) + 1
            view, seg = self._stage(arr, n, dtype, None)
            assert view.numel() == n and np.array_equal(view.numpy(), arr), (n, self.pos)
            self._finish(seg, None)
            assert 0 <= self.pos < self.words                 # never parked on the end of the ring
            return seg

    r = Ring()
    assert r.put(10) == 0 and r.pos == 10 and r.log == []
    assert r.put(6) == 0                                   # exact fill of segment 0: turned at once
    assert r.pos == 16 and r.log == [('record', 0), ('await', 1)]
    assert r.put(16, torch.float32) == 1 and r.pos == 32   # a whole segment in one upload
    assert r.log[-2:] == [('record', 1), ('await', 2)]
    r.put(9)
    assert r.put(9) == 3 and r.pos == 48 + 9               # does not fit the rest of segment 2: segment 3
    assert r.log[-2:] == [('record', 2), ('await', 3)]
    assert r.put(7) == 3                                   # exact fill of the LAST segment: back to word 0, not parked on 64
    assert r.pos == 0 and r.log[-2:] == [('record', 3), ('await', 0)]
    for lap in range(3):                                   # many laps of mixed sizes incl. exact fills: never an empty slice,
        for n in (1, 15, 16, 3, 13, 8, 8, 5, 11, 16, 2, 2, 12):
            r.put(n)                                       # every byte written where the view says
    records = [s for k, s in r.log if k == 'record']
    awaits = [s for k, s in r.log if k == 'await']
    assert awaits == [(s + 1) % 4 for s in records]        # every segment entered was awaited, every segment left recorded
    assert all(b == (a + 1) % 4 for a, b in zip(records, records[1:]))     # segments are left in ring order, none skipped


def test_point_state_merge_keeps_the_newest_refinement():
    """the per-rank point dictionaries of a checkpoint are merged by refinement stamp, not by rank order."""
    from point_teacher_amd.runner import merge_point_states
    t = torch.tensor
    r0 = dict(count=20, epoch=1, gt_bboxes_point={'a': t([1.]), 'b': t([2.])}, refined_gt_bboxes_point={'a': t([10.]), 'b': t([20.])},
              point_stamp={'a': 17, 'b': 3})
    r1 = dict(count=20, epoch=1, gt_bboxes_point={'b': t([2.5]), 'c': t([3.])}, refined_gt_bboxes_point={'b': t([25.]), 'c': t([30.])},
              point_stamp={'b': 12, 'c': 5})
    r2 = dict(count=20, epoch=1, gt_bboxes_point={'a': t([1.5]), 'd': t([4.])}, refined_gt_bboxes_point={'a': t([15.])}, point_stamp={'a': 9})
    m = merge_point_states([r0, r1, r2])
    assert m['count'] == 20 and set(m['refined_gt_bboxes_point']) == {'a', 'b', 'c'} and set(m['gt_bboxes_point']) == {'a', 'b', 'c', 'd'}
    assert float(m['refined_gt_bboxes_point']['a']) == 10. and float(m['gt_bboxes_point']['a']) == 1.       # rank 0 refined it last (17 > 9)
    assert float(m['refined_gt_bboxes_point']['b']) == 25. and float(m['gt_bboxes_point']['b']) == 2.5      # rank 1 (12 > 3): dict.update order would agree ...
    m2 = merge_point_states([r1, r0, r2])
    assert float(m2['refined_gt_bboxes_point']['b']) == 25. and float(m2['refined_gt_bboxes_point']['a']) == 10.   # ... but no longer decides
    assert m['point_stamp'] == {'a': 17, 'b': 12, 'c': 5} and float(m['gt_bboxes_point']['d']) == 4.


def test_epoch_batches_report_the_epoch_a_resume_must_start_with():
    """after the last batch of loader epoch e an end-of-epoch checkpoint must store e + 1."""
    from point_teacher_amd.datasets.loader import EpochBatches

    class Loader:
        def __init__(self):
            self.epochs = []

        def set_epoch(self, e):
            self.epochs.append(e)

        def __len__(self):
            return 3

        def __iter__(self):
            e = self.epochs[-1]
            return iter([(e, i) for i in range(3)])
    b = EpochBatches(Loader())
    assert b.next_epoch() == 0
    assert [b() for _ in range(2)] == [(0, 0), (0, 1)] and b.next_epoch() == 0          # mid-epoch: the same epoch again
    assert b() == (0, 2) and b.epoch == 0 and b.next_epoch() == 1                       # used up: the next one
    resumed = EpochBatches(Loader(), start_epoch=b.next_epoch())
    assert resumed() == b() == (1, 0)                                                   # the resumed run and the uninterrupted one agree


def test_bench_matrix_family_accounting():
    """bench.py prices the matrix family by the EXECUTED bf16 FLOPs: 2 M N K per GEMM launch, 2 M Cout taps Cin per convolution
    launch - forward / input gradient and the weight gradient alike; the operand-format passes are its companions."""
    import bench
    assert bench.algorithmic_flops('pt_gemm_bf16x6_nt', dict(M=5000, N=1024, K=12544)) == 2.0 * 5000 * 1024 * 12544
    conv = dict(P=2 * 100 * 100, Cin=256, Cout=256)
    assert bench.algorithmic_flops('pt_conv3x3_bf16x6_nhwc', conv) == 2.0 * 20000 * 256 * 9 * 256
    assert bench.algorithmic_flops('pt_conv3x3_wgrad_bf16x6_nhwc', conv) == bench.algorithmic_flops('pt_conv3x3_bf16x6_nhwc', conv)
    assert bench.algorithmic_flops('pt_roi_align_fwd', dict(K=1)) is None
    assert set(bench.MFMA_FAMILY[1]) == {'pt_conv_bf16x6', 'pt_conv_wgrad_bf16x6', 'pt_gemm_bf16x6_nt', 'pt_conv3x3_bf16x6_nhwc',
                                         'pt_conv3x3_wgrad_bf16x6_nhwc'}
    import point_teacher_amd.hip as hip
    # the plane-native entry points carry their shapes in a host descriptor: M = B * Ho * Wo output pixels, taps = KH * KW
    d = hip.STRUCTS['pt_conv_desc']()
    d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = 6, 100, 100, 512, 256, 1, 1, 2, 0
    shp = bench.conv_desc_shapes(d)
    assert {k: shp[k] for k in ('M', 'Cin', 'Cout', 'taps', 'np', 'f16')} == dict(M=6 * 50 * 50, Cin=512, Cout=256, taps=1, np=3, f16=0)   # (np unset: three planes)
    # algorithmic bytes = every tensor once as fp32 (a strided 1x1 reads only the pixels it needs), format bytes = the same in 6-byte planes
    assert shp['abytes'] == 4 * (15000 * 512 + 256 * 512) and shp['fbytes'] == 6 * (15000 * 512 + 256 * 512)
    d.out_planes, d.out_f16, d.operand_f16, d.res_planes, d.res_f16, d.mask_planes = 1, 1, 1, 1, 1, 1
    s2 = bench.conv_desc_shapes(d)                                               # H2 operands, H2 result + identity, a ReLU mask
    assert s2['abytes'] == 4 * (15000 * 512 + 256 * 512 + 3 * 15000 * 256)
    assert s2['fbytes'] == 4 * (15000 * 512 + 256 * 512 + 2 * 15000 * 256) + 2 * 15000 * 256
    d.out_planes = d.out_f16 = d.operand_f16 = d.res_planes = d.res_f16 = d.mask_planes = 0
    assert bench.executed_flops('pt_conv_bf16x6', shp) == 6.0 * bench.algorithmic_flops('pt_conv_bf16x6', shp)
    d.np = 1                                                                     # bf16 operands (BASELINE configs[2]): one MFMA product
    assert bench.executed_flops('pt_conv_bf16x6', bench.conv_desc_shapes(d)) == bench.algorithmic_flops('pt_conv_bf16x6', shp)
    d.np = 3
    d.operand_f16 = 1                                                            # fp16 x 2 operands: three products
    assert bench.executed_flops('pt_conv_bf16x6', bench.conv_desc_shapes(d)) == 3.0 * bench.algorithmic_flops('pt_conv_bf16x6', shp)
    d.operand_f16 = 0
    assert bench.algorithmic_flops('pt_conv_bf16x6', shp) == bench.algorithmic_flops('pt_conv_wgrad_bf16x6', shp) == 2.0 * 15000 * 256 * 512
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    assert bench.conv_desc_shapes(d)['M'] == 60000 and bench.conv_desc_shapes(d)['taps'] == 9
    for fn in bench.MFMA_FAMILY[1] + bench.SPLIT_FNS + tuple(m for ms in bench.FAMILIES.values() for m in ms):
        assert fn in hip.PROTOS, fn                                # every entry point the roofline names exists in the header


def test_black_paper_batches_beyond_one_segmented_launch_are_grouped():
    """samples_per_gpu > 16, an image with more than 4 091 objects (2 G + 10 > 8192 candidates) or mixed image shapes: the batch is
    cut into groups one segmented NMS / rasteriser launch takes instead of raising PT_ELIMIT in the middle of training."""
    from point_teacher_amd.proposals import SEGMENT_MAX_CANDIDATES, SEGMENT_MAX_IMAGES, plan_black_paper_groups
    assert (SEGMENT_MAX_IMAGES, SEGMENT_MAX_CANDIDATES) == (16, 8192)
    sh = ((3, 800, 800), 'f32')
    assert plan_black_paper_groups([300] * 2, [sh] * 2) == [(0, 2)]
    assert plan_black_paper_groups([300] * 40, [sh] * 40) == [(0, 16), (16, 32), (32, 40)]
    assert plan_black_paper_groups([300, 5000, 300, 300], [sh] * 4) == [(0, 1), (1, 2), (2, 4)]        # 2 * 5000 + 10 candidates: alone
    other = ((3, 640, 800), 'f32')
    assert plan_black_paper_groups([10, 10, 10], [sh, other, other]) == [(0, 1), (1, 3)]
    assert plan_black_paper_groups([], []) == []
    for counts in ([7] * 33, [4091, 4092, 1], [1]):
        g = plan_black_paper_groups(counts, [sh] * len(counts))
        assert [i for a, b in g for i in range(a, b)] == list(range(len(counts)))                       # a partition, in order
        assert all(b - a <= 16 and (b - a == 1 or max(2 * c + 10 for c in counts[a:b]) <= 8192) for a, b in g)


def test_f16_census_policy_digest():
    """planes.Census turns the device counters into fall-back decisions on the host (no GPU needed for the rule itself): a site that
    saturated demotes its group; so does a non-zero tensor whose largest STORED magnitude sank below 0.25; an all-zero tensor (a masked
    gradient, a pass that did not run) and a healthy one do nothing; '<group>_grad' demotions switch chains to per-link re-scaling."""
    import struct
    import warnings
    from point_teacher_amd import planes as PL
    c = PL.Census()
    c.mode = 2
    c.names = [('trunk', 'layer2.0:y1'), ('towers', 'conv:y'), ('fc', 'fc1:y'), ('trunk_grad', 'layer3.1:E2'), ('neck', 'zero')]
    c.index = {k: i for i, k in enumerate(c.names)}
    bits = lambda v: struct.unpack('<i', struct.pack('<f', v))[0]       # noqa: E731
    host = torch.zeros((8, 4), dtype=torch.int32)
    host[0] = torch.tensor([0, bits(37.5), 10, 1000])                    # healthy
    host[1] = torch.tensor([3, bits(60000.0), 0, 1000])                  # saturated
    host[2] = torch.tensor([0, bits(0.1), 900, 1000])                    # sank below the floor
    host[3] = torch.tensor([0, bits(20000.0), 0, 1000])                  # a chain that grew, still inside the range
    host[4] = torch.tensor([0, 0, 0, 1000])                              # all zeros: no information
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out = c._digest(host)
    assert set(c.demoted) == {'towers', 'fc'} and len(w) == 2
    assert out[('towers', 'conv:y')]['saturated'] == 3 and abs(out[('fc', 'fc1:y')]['amax_stored'] - 0.1) < 1e-7
    assert ('neck', 'zero') in out and c.ok('trunk') and c.ok('trunk_grad') and not c.ok('fc')
    lv = c.log_vars()
    assert lv['census/towers/conv:y/sat'] == 3.0 and lv['census/fc/fc1:y/tiny_frac'] == 0.9 and lv['census/trunk/layer2.0:y1/amax'] == 37.5
    assert c.history[('trunk_grad', 'layer3.1:E2')][1] == 20000.0
    # a saturating gradient site demotes the '<group>_grad' policy only
    host.zero_()
    host[3] = torch.tensor([5, bits(60000.0), 0, 1000])
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        c._digest(host)
    assert 'trunk_grad' in c.demoted and 'trunk' not in c.demoted
    # PT_F16_FALLBACK=0: the census still reports, nothing is demoted
    c2 = PL.Census()
    c2.auto_fallback = False
    c2.names, c2.index = list(c.names), dict(c.index)
    host[1] = torch.tensor([3, bits(60000.0), 0, 1000])
    assert c2._digest(host)[('towers', 'conv:y')]['saturated'] == 3 and not c2.demoted
    c.reset()
    assert not c.demoted and not c.history


def test_compact_alive_batch_matches_per_image():
    """detectors._compact_alive_batch (the batch's fixed-size synthetic lists in one pass) = _compact_alive image by image."""
    from point_teacher_amd.detectors import TS_P2B_FCOS
    g = torch.Generator().manual_seed(11)
    for B, N, n in ((2, 37, 20), (3, 8, 20), (1, 50, 5)):
        boxes = torch.rand(B, N, 4, generator=g) * 100
        alive = torch.rand(B, N, generator=g) > 0.4
        bt, bw = TS_P2B_FCOS._compact_alive_batch(None, boxes, alive, n)
        for i in range(B):
            t, w = TS_P2B_FCOS._compact_alive(None, boxes[i], alive[i], n)
            assert torch.equal(bt[i], t) and torch.equal(bw[i], w)
