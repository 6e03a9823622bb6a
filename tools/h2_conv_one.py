"""One plane convolution on H2 (scaled fp16 x 2) operands a few times, for rocprofv3 --pmc passes:
python tools/h2_conv_one.py B H W Cin Cout K [reps] [b3]      (FC1: 1 5000 1 12544 1024 1; tower: 2 100 100 256 256 3)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F, planes as PL  # noqa: E402

B, H, W, Cin, Cout, K = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 and sys.argv[7].isdigit() else 5
h2 = 'b3' not in sys.argv[1:]
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, K, K, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
sh = torch.randn(Cout, generator=g).to(dev)
xt = PL.split_nhwc(x, f16=h2)[0]
wp = F._conv_weight_planes(w, False, None, 2 if h2 else 3)
for _ in range(reps):
    PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, (K - 1) // 2, shift=sh, relu=True, want_planes=True, splits=1)
torch.cuda.synchronize()
