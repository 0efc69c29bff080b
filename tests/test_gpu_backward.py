"""GPU parity of the backward kernels (SPEC.md §16) (-m gpu).

Scatter-add tolerance: |gpu - oracle| <= 1e-5 * sum|terms| per element (float atomics add in an
unspecified order; the oracle sums in binary64).  max_pool_s and its backward are exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _scatter_ref(gout, idx, N):
    B, C, M, S = gout.shape
    ref = np.zeros((B, C, N), np.float64)
    mag = np.zeros((B, C, N), np.float64)
    for b in range(B):
        j = idx[b].reshape(-1)
        for c in range(C):
            np.add.at(ref[b, c], j, gout[b, c].reshape(-1).astype(np.float64))
            np.add.at(mag[b, c], j, np.abs(gout[b, c].reshape(-1)).astype(np.float64))
    return ref, mag


@pytest.mark.parametrize("B,C,N,M,S", [(2, 5, 300, 40, 8), (1, 64, 1024, 256, 32), (2, 3, 50, 7, 16), (1, 17, 4096, 512, 64)])
def test_group_points_grad(orc, sad, dev, B, C, N, M, S):
    from sad_amd import autograd as ag
    rng = np.random.default_rng(B + C + N)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    idx = orc.ball_query(0.2, S, xyz, np.ascontiguousarray(xyz[:, :M]))     # realistic: padded groups, heavy collisions
    gout = rng.standard_normal((B, C, M, S)).astype(np.float32)
    ref, mag = _scatter_ref(gout, idx, N)
    for via_pm in (True, False):        # point-major atomics + transpose (default) / direct channel-major scatter
        got = ag.group_points_grad(_t(gout, dev), _t(idx, dev), N, via_point_major=via_pm).cpu().numpy()
        assert np.all(np.abs(got - ref) <= 1e-5 * mag + 1e-30), f"via_point_major={via_pm}"
    got_pm = ag.group_points_grad(_t(gout, dev), _t(idx, dev), N, point_major=True).cpu().numpy()
    assert np.all(np.abs(got_pm.transpose(0, 2, 1) - ref) <= 1e-5 * mag + 1e-30)
    # gather_points_grad = the S == 1 case
    fidx = orc.fps(xyz, M)
    g1 = rng.standard_normal((B, C, M)).astype(np.float32)
    got1 = ag.group_points_grad(_t(g1, dev), _t(fidx, dev), N).cpu().numpy()
    ref1, mag1 = _scatter_ref(g1[..., None], fidx[..., None], N)
    assert np.all(np.abs(got1 - ref1) <= 1e-5 * mag1 + 1e-30)


@pytest.mark.parametrize("B,C,M,S", [(2, 7, 33, 16), (1, 128, 512, 32), (1, 3, 5, 1), (2, 16, 100, 24)])
def test_max_pool_s_exact(sad, dev, B, C, M, S):
    import torch
    from sad_amd import autograd as ag
    rng = np.random.default_rng(S + M)
    x = rng.integers(-3, 4, (B, C, M, S)).astype(np.float32)      # small integers: many exact ties
    out, arg = ag.max_pool_s_with_arg(_t(x, dev))
    np.testing.assert_array_equal(out.cpu().numpy(), x.max(-1))
    np.testing.assert_array_equal(arg.cpu().numpy(), x.argmax(-1).astype(np.int32))   # numpy: first maximum
    g = rng.standard_normal((B, C, M)).astype(np.float32)
    xt = _t(x, dev).requires_grad_(True)
    ag.max_pool_s(xt).backward(_t(g, dev))
    want = np.zeros_like(x)
    np.put_along_axis(want, x.argmax(-1)[..., None], g[..., None], axis=-1)
    np.testing.assert_array_equal(xt.grad.cpu().numpy(), want)


def test_unfused_sa_stack_trains(orc, sad, dev):
    """group -> 1x1 conv (torch) -> ReLU -> max over nsample, differentiated through this package's
    kernels, against the same stack written with torch indexing ops."""
    import torch
    from sad_amd import autograd as ag, ops
    rng = np.random.default_rng(16)
    B, C, N, M, S, Co = 2, 6, 500, 60, 16, 10
    xyz = rng.random((B, N, 3), dtype=np.float32)
    idx = _t(orc.ball_query(0.25, S, xyz, np.ascontiguousarray(xyz[:, :M])), dev)
    f0 = rng.standard_normal((B, C, N)).astype(np.float32)
    w0 = (rng.standard_normal((Co, C)) * 0.3).astype(np.float32)

    def run(use_ours):
        feat = _t(f0, dev).requires_grad_(True)
        w = _t(w0, dev).requires_grad_(True)
        if use_ours:
            g = ag.group_points(feat, idx)
        else:
            li = idx.long().reshape(B, 1, M * S).expand(B, C, M * S)
            g = torch.gather(feat, 2, li).reshape(B, C, M, S)
        y = torch.relu(torch.einsum("oc,bcms->boms", w, g))
        p = ag.max_pool_s(y) if use_ours else y.max(dim=3).values
        loss = (p * p).sum()
        loss.backward()
        return loss.item(), feat.grad.cpu().numpy(), w.grad.cpu().numpy()

    l1, gf1, gw1 = run(True)
    l2, gf2, gw2 = run(False)
    assert abs(l1 - l2) <= 1e-5 * abs(l2)
    assert np.allclose(gf1, gf2, rtol=1e-4, atol=1e-5) and np.allclose(gw1, gw2, rtol=1e-4, atol=1e-4)
