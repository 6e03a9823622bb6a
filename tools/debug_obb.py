"""Side-by-side loss dicts of the OBB iteration: HIP path vs oracle (debug aid, GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import test_obb_parity as T  # noqa: E402
from oracle import ref_model as M  # noqa: E402
from oracle import ref_obb as O  # noqa: E402

phase2 = (sys.argv[1] if len(sys.argv) > 1 else 'step2') == 'step2'
dev = torch.device('cuda:0')
pta, cfg, model = T._build(dev, phase2=phase2)
img, boxes, labels, metas = T._data()
g = torch.Generator().manual_seed(11)
neg_u = torch.rand(2, 5, 200, generator=g)
aug = (['horizontal', 'None'], [5, 13], [0.9, 1.1])
model._inject = dict(neg0=neg_u.to(dev), aug=aug)
sd_s0 = T._strip(model.state_dict(), 'student.')
sd_t0 = T._strip(model.state_dict(), 'teacher.')
data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
cap = {}
head = model.student.bbox_head
orig_sel = head.mil_bag_selection


def spy_sel(r, *a, **k):
    out = orig_sel(r, *a, **k)
    cap['merged'] = [o.cpu() for o in out]
    cap['cls'] = r['cls_score'].detach().cpu(); cap['ins'] = r['ins_score'].detach().cpu()
    cap['bags'] = [b.cpu() for b in r['extensive_bags']]
    return out
head.mil_bag_selection = spy_sel
orig_pb = model.teacher.bbox_head.get_pseudo_bbox


def spy_pb(*a, **k):
    out = orig_pb(*a, **k)
    cap['pb'] = [o.cpu() for o in out[0]]
    return out
model.teacher.bbox_head.get_pseudo_bbox = spy_pb
out = model.train_step(data, None)
lv = out['log_vars'].materialize()
torch.set_num_threads(8)
params = {k: v for k, v in sd_s0.items()}
sd_t = M.ema(sd_t0, sd_s0)
gp = [b[:, :2] for b in boxes]
# oracle with capture
orig = O.mil_bag_select_obb


def spy_o(cls, ins, valid, labels_, bags, pseudo, *a, **k):
    cap['o_cls'], cap['o_ins'], cap['o_bags'], cap['o_pseudo'] = cls.detach(), ins.detach(), bags, pseudo
    r = orig(cls, ins, valid, labels_, bags, pseudo, *a, **k)
    cap['o_merged'] = r
    return r
O.mil_bag_select_obb = spy_o
with torch.no_grad():
    ref, _ = O.forward_train_step2(params, sd_t, img, boxes, labels, gp, dict(O.MODEL_CFG), dict(neg0=neg_u, aug=aug))
ref['loss'] = M.total_loss(ref)
for k in ref:
    a, b = float(lv[k]), float(ref[k])
    print(f'{k:32s} hip {a:12.6f}  oracle {b:12.6f}  rel {abs(a - b) / max(abs(b), 1e-2):.2e}')
print('pseudo boxes max diff', float((torch.cat(cap['pb']) - cap['o_pseudo']).abs().max()))
print('bags max diff', float((torch.cat(cap['bags']) - cap['o_bags']).abs().max()))
print('cls max diff', float((cap['cls'] - cap['o_cls']).abs().max()), 'ins', float((cap['ins'] - cap['o_ins']).abs().max()))
d = (torch.cat(cap['merged']) - cap['o_merged']).abs()
print('merged max diff', float(d.max()), 'rows >1e-2:', d.max(1)[0].gt(1e-2).nonzero().reshape(-1).tolist())
import torch.nn.functional as TF
for name, cls_, ins_ in (('hip', cap['cls'], cap['ins']), ('oracle', cap['o_cls'], cap['o_ins'])):
    N, U1, U2, C = cls_.shape
    lab = torch.cat(labels)
    c = cls_.reshape(N, U1 * U2, C).sigmoid()
    i = TF.normalize(ins_.softmax(2), dim=2, p=1).reshape(N, U1 * U2, C)
    s = c[torch.arange(N), :, lab] * i[torch.arange(N), :, lab]
    for r in d.max(1)[0].gt(1e-2).nonzero().reshape(-1).tolist():
        v, ix = s[r].topk(5)
        print(name, 'row', r, 'top5', [f'{float(x):.6e}' for x in v], ix.tolist())
