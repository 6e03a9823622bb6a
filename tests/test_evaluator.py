"""AI-TOD / COCO-style evaluator (row N1): the numpy oracle against hand-checkable answers (CPU), the GPU evaluator
(pt_coco_match + batched accumulation) against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import ref_cocoeval as RC


def _scene(seed, n_img=6, K=3, with_flags=True):
    rng = np.random.RandomState(seed)
    gts, res = [], []
    for i in range(n_img):
        G = rng.randint(0, 14)
        c = rng.rand(G, 2) * 300 + 20
        wh = np.exp(rng.randn(G, 2) * 0.7 + np.log(12.0)).clip(2, 90)           # spans verytiny .. medium
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        lab = rng.randint(0, K, G)
        g = dict(bboxes=b, labels=lab)
        if with_flags:
            g['iscrowd'] = (rng.rand(G) < 0.1).astype(int)
            g['ignore'] = (rng.rand(G) < 0.1).astype(int)
        gts.append(g)
        per = []
        for k in range(K):
            gb = b[lab == k]
            keep = rng.rand(len(gb)) < 0.8                                       # some misses
            d = gb[keep] + rng.randn(int(keep.sum()), 4).astype(np.float32) * 1.5   # jittered true positives
            d = np.concatenate([d, d[: len(d) // 3] + 0.7], 0)                   # duplicates
            nf = rng.randint(0, 5)
            fc = rng.rand(nf, 2) * 300 + 20
            fw = np.exp(rng.randn(nf, 2) * 0.7 + np.log(12.0)).clip(2, 90)
            d = np.concatenate([d, np.concatenate([fc - fw / 2, fc + fw / 2], 1)], 0).astype(np.float32)
            s = rng.rand(len(d), 1).astype(np.float32)
            per.append(np.concatenate([d, s], 1))
        res.append(per)
    return gts, res


def test_oracle_known_answers():
    """Perfect detections -> every metric 1; no detections -> 0; a single false positive ranked first halves AP."""
    gts = [dict(bboxes=np.array([[10, 10, 20, 20], [50, 50, 90, 90]], np.float32), labels=np.array([0, 0]))]
    perfect = [[np.array([[10, 10, 20, 20, .9], [50, 50, 90, 90, .8]], np.float32)]]
    st, pr, rc = RC.evaluate(perfect, gts, 1)
    assert st['mAP'] == pytest.approx(1.0) and st['mAP_50'] == pytest.approx(1.0) and st['AR@1500'] == pytest.approx(1.0)
    assert st['mAP_t'] == pytest.approx(1.0) and st['mAP_m'] == pytest.approx(1.0)       # 10x10 = tiny, 40x40 = medium
    assert st['mAP_vt'] == -1.0 and st['mAP_25'] == -1.0                                  # no such gt / threshold
    st, _, _ = RC.evaluate([[np.zeros((0, 5), np.float32)]], gts, 1)
    assert st['mAP'] == 0.0 and st['AR@1500'] == 0.0
    fp_first = [[np.array([[200, 200, 220, 220, .95], [10, 10, 20, 20, .9], [50, 50, 90, 90, .8]], np.float32)]]
    st, pr, _ = RC.evaluate(fp_first, gts, 1)
    # precision envelope: 1/2 up to recall .5, 2/3 up to recall 1 -> AP = (51 * 2/3 + 50 * 2/3) / 101 = 2/3
    assert st['mAP_50'] == pytest.approx(2 / 3, abs=1e-9)
    # the fork's hard-coded IoU 0.25 (aitod.py:64)
    st, _, _ = RC.evaluate(perfect, gts, 1, iou_thrs=[0.25])
    assert st['mAP'] == pytest.approx(1.0) and st['mAP_25'] == pytest.approx(1.0) and st['mAP_50'] == -1.0


def test_oracle_crowd_and_ignore():
    """A crowd gt absorbs any number of detections without penalty; detections matched to ignored gts do not count."""
    gts = [dict(bboxes=np.array([[10, 10, 30, 30], [100, 100, 160, 160]], np.float32), labels=np.array([0, 0]),
                iscrowd=np.array([0, 1]))]
    dets = [[np.array([[10, 10, 30, 30, .9], [100, 100, 130, 130, .8], [130, 130, 160, 160, .7]], np.float32)]]
    st, _, _ = RC.evaluate(dets, gts, 1)
    assert st['mAP_50'] == pytest.approx(1.0) and st['AR@1500'] == pytest.approx(1.0)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,flags', [(0, True), (1, True), (2, False)])
@pytest.mark.parametrize('thrs', [None, [0.25]])
def test_gpu_evaluator_vs_oracle(seed, flags, thrs):
    from point_teacher_amd.evaluation import AITODEvaluator
    gts, res = _scene(seed, with_flags=flags)
    ref, pr, rc = RC.evaluate(res, gts, 3, iou_thrs=thrs)
    ev = AITODEvaluator(gts, 3, iou_thrs=thrs)
    out = ev.evaluate(res)
    np.testing.assert_allclose(out['precision'].cpu().numpy(), pr, rtol=0, atol=1e-9)
    np.testing.assert_allclose(out['recall'].cpu().numpy(), rc, rtol=0, atol=1e-9)
    for k, v in ref.items():
        assert out['bbox_' + k] == pytest.approx(v, abs=1e-9), k
    assert ref['mAP'] > 0.05                                   # the scene is not degenerate


@pytest.mark.gpu
def test_gpu_evaluator_edge_cases():
    from point_teacher_amd.evaluation import AITODEvaluator
    gts = [dict(bboxes=np.zeros((0, 4), np.float32), labels=np.zeros(0, int)),
           dict(bboxes=np.array([[5, 5, 11, 11]], np.float32), labels=np.array([1]))]
    ev = AITODEvaluator(gts, 2)
    empty = [[np.zeros((0, 5), np.float32)] * 2] * 2
    out = ev.evaluate(empty)
    ref, _, _ = RC.evaluate(empty, gts, 2)
    for k, v in ref.items():
        assert out['bbox_' + k] == pytest.approx(v, abs=1e-12), k
    dets = [[np.array([[1, 1, 9, 9, .5]], np.float32), np.zeros((0, 5), np.float32)],      # detection on an image without gts
            [np.zeros((0, 5), np.float32), np.array([[5, 5, 11, 11, .9]], np.float32)]]
    out = ev.evaluate(dets)
    ref, _, _ = RC.evaluate(dets, gts, 2)
    for k, v in ref.items():
        assert out['bbox_' + k] == pytest.approx(v, abs=1e-12), k
    assert out['bbox_mAP_vt'] == pytest.approx(1.0)            # the 6x6 object is "very tiny" and perfectly found
    # more than maxDets[-1] detections in one (image, class): only the top 1500 are matched
    many = np.concatenate([np.random.RandomState(0).rand(1600, 4).astype(np.float32) * 3 + np.array([5, 5, 8, 8], np.float32),
                           np.linspace(0.99, 0.01, 1600, dtype=np.float32)[:, None]], 1)
    dets = [[np.zeros((0, 5), np.float32)] * 2, [np.zeros((0, 5), np.float32), many]]
    out = ev.evaluate(dets)
    ref, _, _ = RC.evaluate(dets, gts, 2)
    for k, v in ref.items():
        assert out['bbox_' + k] == pytest.approx(v, abs=1e-9), k


@pytest.mark.gpu
def test_eval_loop_end_to_end():
    """single_gpu_test (apis/test.py:16-66) -> AITODEvaluator on synthetic tiles: the loop the reference's tools/test.py
    runs, with the teacher's detections; a detector that returns the ground truth scores 1."""
    import os
    import point_teacher_amd as pta
    from point_teacher_amd.evaluation import AITODEvaluator, single_gpu_test
    from point_teacher_amd.synthetic import SyntheticTiles
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = pta.Config.fromfile(os.path.join(root, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    model = pta.build_detector(cfg.model).to(dev)
    with torch.no_grad():
        model.teacher.bbox_head.conv_cls.bias.fill_(-2.5)
        model.teacher.bbox_head.conv_reg.bias.fill_(1.0)
    data = SyntheticTiles(n=4, size=256, mean_objects=15, seed=2, device=dev)
    results = single_gpu_test(model, lambda it: data.batch(it, 2), 2)
    assert len(results) == 4 and len(results[0]) == 8 and results[0][0].shape[1] == 5
    gts = [dict(bboxes=data.items[i][1].cpu().numpy(), labels=data.items[i][2].cpu().numpy()) for i in range(4)]
    ev = AITODEvaluator(gts, 8)
    out = ev.evaluate(results)
    for k in ('bbox_mAP', 'bbox_mAP_50', 'bbox_mAP_vt', 'bbox_mAP_t', 'bbox_AR@1500'):
        assert -1.0 <= out[k] <= 1.0, (k, out[k])
    ref, _, _ = RC.evaluate(results, gts, 8)
    for k, v in ref.items():
        assert out['bbox_' + k] == pytest.approx(v, abs=1e-6), k
    oracle_det = [[np.concatenate([g['bboxes'][g['labels'] == k], np.full((int((g['labels'] == k).sum()), 1), 0.9, np.float32)], 1)
                   for k in range(8)] for g in gts]
    assert ev.evaluate(oracle_det)['bbox_mAP'] == pytest.approx(1.0)


# ------------------------------------------------------------------ oriented tree: DOTA-style mAP --
def _load_obb_eval():
    from conftest import load_golden
    g = load_golden('obb_eval_map')
    n_img, K = int(g['n_img']), int(g['num_classes'])
    anns = [{key: g[f'in_ann{i}_{key}'] for key in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore')} for i in range(n_img)]
    dets = [[g[f'in_det{i}_{k}'] for k in range(K)] for i in range(n_img)]
    return g, dets, anns, K


@pytest.mark.parametrize('thr', [0.5, 0.25])
def test_oracle_eval_rbbox_map_vs_reference_golden(thr):
    """oracle restatement == the reference's own eval_rbbox_map run (tests/golden/obb_eval_map.npz)."""
    g, dets, anns, K = _load_obb_eval()
    mean_ap, res = RC.eval_rbbox_map(dets, anns, iou_thr=thr)
    t = int(thr * 100)
    assert mean_ap == pytest.approx(float(g[f'out_map_{t}']), abs=1e-6)
    for k in range(K):
        assert res[k]['num_gts'] == int(g[f'out_num_gts_{t}_{k}'])
        assert res[k]['ap'] == pytest.approx(float(g[f'out_ap_{t}_{k}']), abs=1e-6)
        np.testing.assert_allclose(res[k]['recall'], g[f'out_recall_{t}_{k}'], atol=1e-6)
        np.testing.assert_allclose(res[k]['precision'], g[f'out_precision_{t}_{k}'], atol=1e-6)
    assert 0.2 < mean_ap < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('thr', [0.5, 0.25])
def test_gpu_eval_rbbox_map_vs_reference_golden(thr):
    from point_teacher_amd.evaluation import eval_rbbox_map
    g, dets, anns, K = _load_obb_eval()
    mean_ap, res = eval_rbbox_map(dets, anns, iou_thr=thr)
    t = int(thr * 100)
    assert mean_ap == pytest.approx(float(g[f'out_map_{t}']), abs=1e-5)
    for k in range(K):
        assert res[k]['num_gts'] == int(g[f'out_num_gts_{t}_{k}'])
        assert res[k]['ap'] == pytest.approx(float(g[f'out_ap_{t}_{k}']), abs=1e-5)
        np.testing.assert_allclose(res[k]['recall'], g[f'out_recall_{t}_{k}'], atol=1e-6)
        np.testing.assert_allclose(res[k]['precision'], g[f'out_precision_{t}_{k}'], atol=1e-6)
