"""Turn the two --pmc summaries of tools/pmc_pass.sh (FETCH_SIZE and WRITE_SIZE, separate passes as the MI355X guide
prescribes) into profiles/<round>/pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
usage: python tools/pmc_to_json.py <tag> <workload key> [<out.json>]      (reads gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE.txt)

Units and corrections (guide, section HBM): the counters are KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
a wide (16 B / lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte stores and for float atomics.
Narrower loads (the 4-byte gathers of the feature map in RoIAlign, L2-resident) are uncalibrated and are counted as read.
Each entry records the sha256 of the kernel's source file: bench.py reports the number only while that file is unchanged."""
import ast
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {'roi_align7_fwd': ('pt_roi_align', 'roi_align.hip'), 'roi_align7_bwd': ('pt_roi_align', 'roi_align.hip'),
          'roi_align_fwd_cl': ('pt_roi_align', 'roi_align.hip'), 'roi_align_bwd_cl': ('pt_roi_align', 'roi_align.hip'),
          'affine_relu_fwd_kernel': ('pt_affine_relu', 'optim.hip'), 'affine_relu_bwd_kernel': ('pt_affine_relu', 'optim.hip'),
          'affine_relu_fwd_bf16_kernel': ('pt_affine_relu', 'optim.hip'), 'affine_relu_bwd_bf16_kernel': ('pt_affine_relu', 'optim.hip'),
          'ema_kernel': ('pt_optimizer (ema + sqnorm + sgd)', 'optim.hip'), 'sgd_kernel': ('pt_optimizer (ema + sqnorm + sgd)', 'optim.hip'),
          'sqnorm_kernel': ('pt_optimizer (ema + sqnorm + sgd)', 'optim.hip'),
          'gemm_bf16x6_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip'),
          'wgrad_bf16x6_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip'),
          # second halves of the same entry-point launches (split-k sum + epilogue, fixed-order sum of the pixel chunks): their bytes
          # belong to the launch, they are not launches of their own
          'conv_splitk_finish_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip', False),
          'wgrad_reduce_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip', False),
          'bn_wgrad_finish_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip', False),
          'planes_mix_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip', False),
          'gather_h2_kernel': ('pt_bf16x6 (conv + gemm + wgrad)', 'gemm_split.hip', False),
          'roi_align_rotated_fwd_mm': ('pt_roi_align_rotated', 'rotated.hip'), 'roi_align_rotated_bwd_mm': ('pt_roi_align_rotated', 'rotated.hip')}


def read(path):
    out = {}
    for line in open(path):
        m = re.match(r'pt::(\w+)(?:<[^>]*>)? (\{.*\})', line.strip())
        if m:
            d = ast.literal_eval(m.group(2))
            (mean, n), = d.values()
            k = out.setdefault(m.group(1), [0.0, 0])
            k[0] += mean * n; k[1] += n
    return out


tag, key = sys.argv[1], sys.argv[2]
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'profiles', 'r05', 'pmc_traffic.json')
fetch = read(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_FETCH_SIZE.txt'))
write = read(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_WRITE_SIZE.txt'))
fam = {}
for k in fetch:
    if k not in FAMILY:
        continue
    name, src = FAMILY[k][:2]
    f = fam.setdefault(name, dict(bytes=0.0, launches=0, kernels={}, source=src))
    n = fetch[k][1]
    byts = (2.0 * fetch[k][0] + write.get(k, [0.0, 0])[0] * (n / max(write.get(k, [0, n])[1], 1))) * 1024
    f['bytes'] += byts
    if len(FAMILY[k]) < 3:
        f['launches'] += n
    f['kernels'][k] = dict(launches=n, fetch_kib_mean=round(fetch[k][0] / n, 1),
                           write_kib_mean=round(write.get(k, [0.0, 1])[0] / max(write.get(k, [0, 1])[1], 1), 1))
doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
doc['_doc'] = __doc__.split('usage')[0].strip() + ' FETCH_SIZE x2, WRITE_SIZE x1, KiB -> bytes.'
doc[key] = {}
for name, f in fam.items():
    sha = hashlib.sha256(open(os.path.join(ROOT, 'point_teacher_amd', 'csrc', f['source']), 'rb').read()).hexdigest()
    doc[key][name] = dict(traffic_bytes_per_launch=int(f['bytes'] / f['launches']), launches_sampled=f['launches'],
                          kernels=f['kernels'], source=f['source'], source_sha256=sha)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(doc, open(out_path, 'w'), indent=1)
print(json.dumps(doc[key], indent=1))
