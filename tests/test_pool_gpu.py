"""GPU: HIP 3x3/s2 max pooling (csrc/pool.hip) against F.max_pool2d on the same bf16 values."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 64, 16, 24), (1, 64, 17, 23), (3, 8, 5, 5), (1, 128, 2, 3)])
def test_maxpool_fwd_bwd(shape):
    from mgnet_amd.modeling import ops

    torch.manual_seed(sum(shape))
    x0 = torch.randn(*shape).to(torch.bfloat16)
    xr = x0.float().requires_grad_(True)
    yr = F.max_pool2d(xr, kernel_size=3, stride=2, padding=1)
    g = torch.randn(*yr.shape).to(torch.bfloat16)
    (yr * g.float()).sum().backward()
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.max_pool_3x3_s2(x)
    (y.float() * g.cuda().float()).sum().backward()
    assert torch.equal(y.float().cpu(), yr.detach())
    assert torch.allclose(x.grad.float().cpu(), xr.grad, atol=2e-2, rtol=1e-2)  # sums of <= 4 bf16 values, rounded to bf16
