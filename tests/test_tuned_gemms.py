"""The shipped GEMM solution table (runtime.enable_tuned_gemms): well-formed on CPU, honoured and numerically neutral on GPU."""
import os

import pytest
import torch

from point_teacher_amd.runtime import TUNED_GEMMS, enable_tuned_gemms


def test_table_is_well_formed():
    rows = [l.strip().split(',') for l in open(TUNED_GEMMS) if l.strip()]
    validators = {r[1]: r[2] for r in rows if r[0] == 'Validator'}
    assert validators['GCN_ARCH_NAME'].startswith('gfx950') and 'PT_VERSION' in validators
    entries = [r for r in rows if r[0] != 'Validator']
    assert all(len(r) == 4 and float(r[3]) > 0 for r in entries)
    shapes = {r[1] for r in entries}
    for k in (5000, 60750):                        # the first FC of the MIL head at the 0 % and 100 % configurations, fwd + dgrad + wgrad
        assert f'tn_1024_{k}_12544_ld_12544_12544_1024' in shapes
        assert f'nn_12544_{k}_1024_ld_12544_1024_12544' in shapes and f'nt_12544_1024_{k}_ld_12544_1024_12544' in shapes
    assert not enable_tuned_gemms() or torch.cuda.is_available()      # never switches anything on without a GPU


@pytest.mark.gpu
def test_table_is_used_and_neutral():
    import torch.cuda.tunable as tunable
    torch.manual_seed(0)
    x = torch.randn(5000, 12544, device='cuda')
    fc = torch.nn.Linear(12544, 1024).cuda()
    tunable.enable(False)
    ref = fc(x)
    assert enable_tuned_gemms()
    assert tunable.is_enabled() and not tunable.tuning_is_enabled()
    assert any('tn_1024_5000_12544' in r[1] for r in tunable.get_results())
    out = fc(x)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)        # another fp32 solution: same product, different summation order
    os.environ['PT_TUNED_GEMMS'] = '0'
    try:
        assert not enable_tuned_gemms()
    finally:
        del os.environ['PT_TUNED_GEMMS']
