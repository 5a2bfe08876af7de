"""siv_sse_fwd_noalloc -- the one entry point that allocates nothing: its scratch is a device array built
into the library.  Called through the C ABI and checked bit for bit against siv_sse_fwd (same kernels, the
partials in a caller workspace) and against the oracle at 1e-4, on two calls in a row, the second with a
smaller input (stale partials of the first call stay in the array and must not be read)."""
import pytest
import torch

from oracle import siv_oracle as O

pytestmark = pytest.mark.gpu


def test_sse_fwd_noalloc_matches_workspace_variant():
    import siv._lib as L
    import siv.backend as sb
    sb.set_backend(None)
    B = sb.get_backend()
    lib = B.lib
    gen = torch.Generator().manual_seed(5)
    stream = torch.cuda.current_stream().cuda_stream
    # 8 full volumes (150 chunks per sample), then 3 samples of a ragged size, then 1 tiny sample
    for shape in ((8, 1, 80, 96, 80), (3, 1, 17, 23, 9), (1, 1, 2, 3, 2)):
        x, y = torch.rand(*shape, generator=gen), torch.rand(*shape, generator=gen)
        xd, yd = x.cuda(), y.cuda()
        want = B.sse_fwd(xd, yd)
        got = torch.full((shape[0],), -1.0, device="cuda")
        before = torch.cuda.memory_allocated()
        L.check(lib.siv_sse_fwd_noalloc(xd.data_ptr(), yd.data_ptr(), got.data_ptr(), shape[0],
                                        x.numel() // shape[0], stream), "siv_sse_fwd_noalloc")
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        assert torch.equal(got, want), (shape, got, want)
        ref = O.calc_reconstruction_loss(x, y, "none")
        assert float((got.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # beyond the built-in scratch: a clear error, nothing launched
    n_big = 257
    xb = torch.zeros(n_big, 4, device="cuda")
    ob = torch.full((n_big,), -1.0, device="cuda")
    rc = lib.siv_sse_fwd_noalloc(xb.data_ptr(), xb.data_ptr(), ob.data_ptr(), n_big, 4, stream)
    assert rc == -1 and b"256" in lib.siv_last_error()
    torch.cuda.synchronize()
    assert float(ob.min()) == -1.0
    with pytest.raises(L.SivError):
        L.check(lib.siv_sse_fwd_noalloc(None, None, None, 1, 4, stream), "siv_sse_fwd_noalloc")
