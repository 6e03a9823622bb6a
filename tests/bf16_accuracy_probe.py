"""Probe behind profiles/r02/bf16_accuracy.txt (run on the GPU box: python tests/bf16_accuracy_probe.py): the bf16 backbone of
configs[2] against the bf16-rounding oracle, layouts, run-to-run reproducibility and MIOpen settings.  Not a test (no test_ prefix)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_train_step_parity as T
from oracle import ref_model as M
dev = torch.device('cuda:0')
pta, cfg, model = T._build(dev, phase2=True)
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, autocast_dtype=torch.bfloat16)
img, boxes, labels, metas = T._data(dev, seed=5)
sd = T._strip(model.state_dict(), 'teacher.')
feats = {}
def hook(name):
    def f(m, i, o): feats[name] = (o[0] if isinstance(o, (tuple, list)) else o).detach().float().cpu()
    return f
t = model.teacher
hs = [t.backbone.conv1.register_forward_hook(hook('conv1')), t.backbone.layer1.register_forward_hook(hook('layer1')),
      t.backbone.layer2.register_forward_hook(hook('layer2')), t.backbone.layer3.register_forward_hook(hook('layer3')),
      t.backbone.layer4.register_forward_hook(hook('layer4')), t.neck.register_forward_hook(hook('fpn0')), t.neck_agg.register_forward_hook(hook('psagg'))]
with torch.no_grad():
    out = t.extract_feat(img.to(dev))[0].float().cpu()
    for cl in (False, True):
        x = img.to(dev)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        o2 = t.extract_feat(x)[0].float().cpu()
        print('channels_last', cl, 'vs first run', float((o2-out).norm()/out.norm()))
# oracle with taps
import torch.nn.functional as F
q = lambda x: x.bfloat16().float()
with torch.no_grad():
    with M.bf16_backbone():
        ref = M.extract_feat(sd, img)
    ref32 = M.extract_feat(sd, img)
    c1 = q(F.conv2d(q(img), q(sd['backbone.conv1.weight']), None, 2, 3))
print('conv1 product vs oracle', float((feats['conv1']-c1).norm()/c1.norm()), 'max', float((feats['conv1']-c1).abs().max()), float(c1.abs().max()))
print('final: product vs bf16 oracle', float((out-ref).norm()/ref.norm()), ' fp32 oracle vs bf16 oracle', float((ref32-ref).norm()/ref.norm()), ' product vs fp32 oracle', float((out-ref32).norm()/ref32.norm()))
# fp32 product
for m in model.modules():
    if hasattr(m, 'backbone') and hasattr(m, 'extract_feat'): m.backbone_autocast = None
with torch.no_grad():
    o32 = t.extract_feat(img.to(dev))[0].float().cpu()
print('fp32 product vs fp32 oracle', float((o32-ref32).norm()/ref32.norm()))
for k in ('layer1','layer2','layer3','layer4'):
    print(k, feats[k].shape, float(feats[k].abs().mean()))
print('---- settings sweep (bf16 teacher features vs bf16 oracle; repeatability)')
for m in model.modules():
    if hasattr(m, 'backbone') and hasattr(m, 'extract_feat'): m.backbone_autocast = torch.bfloat16
import time
for bench in (False, True):
    for det in (False, True):
        torch.backends.cudnn.benchmark = bench
        torch.backends.cudnn.deterministic = det
        x = img.to(dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = t.extract_feat(x)[0].float().cpu()
            b = t.extract_feat(x)[0].float().cpu()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(5): t.extract_feat(x)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print(f'benchmark={bench} deterministic={det}: vs bf16 oracle {float((a-ref).norm()/ref.norm()):.4f}  run-to-run {float((a-b).norm()/a.norm()):.2e}  {dt*1e3:.2f} ms/pass (256x256, B=2)')
