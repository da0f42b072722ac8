"""stmi_instance_norm_nhwc_stats_static -- the entry point that allocates nothing -- against
stmi_instance_norm_nhwc_stats, through the C-ABI (ctypes), bit for bit: two calls in a row, the
second on a smaller input (the fixed workspace keeps the first call's contents), then the size
limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stats(entry, x, eps, with_ws):
    from stmi import lib
    n, h, w, c = x.shape
    stats = torch.full((n, c, 2), float('nan'), dtype=torch.float32, device=x.device)
    dt = lib.BF16 if x.dtype == torch.bfloat16 else lib.F32
    stream = torch.cuda.current_stream().cuda_stream
    if with_ws:
        ws = torch.empty(lib.load().stmi_instance_norm_ws_floats(n, h, w, c), dtype=torch.float32, device=x.device)
        lib.call(entry, x.data_ptr(), ws.data_ptr(), stats.data_ptr(), eps, n, h, w, c, dt, stream)
    else:
        lib.call(entry, x.data_ptr(), stats.data_ptr(), eps, n, h, w, c, dt, stream)
    torch.cuda.synchronize()
    return stats


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_static_workspace_stats_match_the_workspace_entry_point(cuda, dtype):
    from stmi import lib
    g = torch.Generator(device='cpu').manual_seed(3)
    # large first (many partial rows per sample), then smaller in every dimension
    for shape in ((4, 96, 80, 128), (2, 17, 23, 32), (1, 5, 3, 3)):
        x = (torch.randn(shape, generator=g) * 3 + 1.5).to(cuda, dtype).contiguous()
        want = _stats('stmi_instance_norm_nhwc_stats', x, 1e-3, True)
        got = _stats('stmi_instance_norm_nhwc_stats_static', x, 1e-3, False)
        assert torch.isfinite(want).all()
        assert torch.equal(got, want), shape
        # and the numbers are the statistics: mean and 1/sqrt(var + eps) over H*W
        xf = x.double()
        mean = xf.mean(dim=(1, 2))
        rstd = 1.0 / torch.sqrt(xf.var(dim=(1, 2), unbiased=False) + 1e-3)
        assert torch.allclose(got[..., 0].double(), mean, rtol=1e-4, atol=1e-4)
        assert torch.allclose(got[..., 1].double(), rstd, rtol=1e-3, atol=1e-5)
    # beyond the fixed workspace: refused with a message, nothing launched
    h = lib.load()
    big = torch.zeros((1, 1, 1, 8), dtype=dtype, device=cuda)
    stats = torch.zeros((8193, 1, 2), dtype=torch.float32, device=cuda)
    rc = h.stmi_instance_norm_nhwc_stats_static(big.data_ptr(), stats.data_ptr(), 1e-3, 8193, 1, 1, 1,
                                                lib.BF16 if dtype == torch.bfloat16 else lib.F32, None)
    assert rc == -1 and b'exceeds the fixed workspace' in h.stmi_last_error()
