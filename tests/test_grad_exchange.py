"""The device side of the sparse gradient exchange (radfoam_amd/csrc/rf_grad_exchange.hip) against the
same steps written with torch indexing (the CPU path of SparseGradExchange that the gloo tests run)."""
import numpy as np
import pytest
import torch

from radfoam_amd import dist as rdist


def _case(n, a, density, seed, device):
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(n * (3 + a), dtype=torch.float32)
    pg, ag = flat[: 3 * n].view(n, 3), flat[3 * n:].view(n, a)
    rows = (torch.rand(n, generator=g) < density).nonzero().reshape(-1)
    ag[rows] = torch.randn(len(rows), a, generator=g)
    pg[rows[::2]] = torch.randn(len(rows[::2]), 3, generator=g)
    only_pg = rows[1::7]                       # rows whose only non-zero values are point gradients
    ag[only_pg] = 0
    pg[only_pg, 1] = 1.5
    ag[rows[3::11], :] = 0                      # ... and a few that end up entirely zero again
    pg[rows[3::11], :] = 0
    pg[rows[3::11][:2], 0] = -0.0               # negative zero alone is not worth a row
    flat = flat.to(device)
    return flat, flat[: 3 * n].view(n, 3), flat[3 * n:].view(n, a)


def _rows_as_dict(send, k, a):
    idx = send[:k, 0].contiguous().view(torch.int32).tolist()
    assert len(set(idx)) == len(idx)
    return {i: send[j, 1:4 + a].clone() for j, i in enumerate(idx)}


@pytest.mark.gpu
@pytest.mark.parametrize("a", [4, 13, 28, 49])
def test_compact_and_scatter_match_the_torch_path(a):
    n = 50_000
    ex = rdist.SparseGradExchange()
    pitch = ex._pitch(a)
    flat_g, pg_g, ag_g = _case(n, a, 0.07, a, "cuda:0")
    flat_c, pg_c, ag_c = _case(n, a, 0.07, a, "cpu")
    cap = 8192
    send_g = torch.full((cap, pitch), 7.0, device="cuda:0")
    send_c = torch.full((cap, pitch), 7.0)
    cnt_g = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    cnt_c = torch.zeros(1, dtype=torch.int32)
    ex._compact(pg_g, ag_g, send_g, cnt_g)
    ex._compact(pg_c, ag_c, send_c, cnt_c)
    k = int(cnt_c)
    assert int(cnt_g) == k and 0 < k <= cap
    got, ref = _rows_as_dict(send_g.cpu(), k, a), _rows_as_dict(send_c, k, a)
    assert got.keys() == ref.keys()
    assert all(torch.equal(got[i], ref[i]) for i in ref)
    assert float(send_g[:k, 4 + a:].abs().sum()) == 0.0          # padding columns are zero
    # capacity overflow: the count is still the true one, nothing is written past the buffer
    small = torch.full((100, pitch), 7.0, device="cuda:0")
    guard = small.clone()
    cnt_g.zero_()
    ex._compact(pg_g, ag_g, small[:64], cnt_g)
    assert int(cnt_g) == k and torch.equal(small[64:], guard[64:])
    # scatter: zero the listed rows, then add them twice -> twice the original, elsewhere untouched
    before = flat_g.clone()
    ex._scatter(send_g, k, pg_g, ag_g, zero=True)
    assert float(flat_g.abs().sum()) == 0.0
    ex._scatter(send_g, k, pg_g, ag_g, zero=False)
    assert torch.equal(flat_g, torch.where(before == 0, torch.zeros_like(before), before))   # -0.0 rows come back as +0
    ex._scatter(send_g, k, pg_g, ag_g, zero=False)
    assert torch.equal(flat_g, 2 * torch.where(before == 0, torch.zeros_like(before), before))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 4099])
@pytest.mark.parametrize("a,pitch", [(13, 16), (28, 32), (49, 64)])
def test_padded_rows_pack_like_dense_rows(n, a, pitch):
    """Pipeline.trace_backward's default layout (rows of `pitch` floats on 64-byte lines, attr_grad returned as a
    [N, A] view) through the exchange's device steps, for N % 16 == 0 and != 0: the same packed rows as from dense rows,
    and a scatter into padded rows leaves the padding columns alone."""
    ex = rdist.SparseGradExchange()
    _, pg, ag = _case(n, a, 0.1, 3 * a + n, "cuda:0")
    padded = torch.zeros((n, pitch), device="cuda:0")
    padded[:, :a] = ag
    view = padded[:, :a]
    assert not view.is_contiguous()
    packs = []
    for rows in (ag, view):
        send = torch.zeros((n, ex._pitch(a)), device="cuda:0")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        ex._compact(pg, rows, send, cnt)
        k = int(cnt)
        packs.append(_rows_as_dict(send.cpu(), k, a))
    assert packs[0].keys() == packs[1].keys() and all(torch.equal(packs[0][i], packs[1][i]) for i in packs[0])
    out_pg, out = torch.zeros_like(pg), torch.full((n, pitch), 5.0, device="cuda:0")
    out[:, :a] = 0
    ex._scatter(send, k, out_pg, out[:, :a], zero=False)
    assert torch.equal(out[:, :a], torch.where(ag == 0, torch.zeros_like(ag), ag))
    assert torch.equal(out_pg, torch.where(pg == 0, torch.zeros_like(pg), pg))
    assert bool((out[:, a:] == 5.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("n_points", [4096, 4100])
def test_sparse_exchange_steps_on_the_pipelines_default_gradient_layout(foam_factory, n_points):
    """... and the real thing: the attr_grad a default ("auto" row pitch) pipeline returns goes through compaction and
    scatter and comes back as the gradients themselves."""
    import radfoam
    from tests import helpers as H
    fm = foam_factory(n_points, 1, 5)
    cam, rays, start = H.camera_setup(fm, 64, 48)
    p, at, adj, off = H.to_torch_foam(fm, "cuda:0")
    r = torch.from_numpy(rays).to("cuda:0")
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to("cuda:0")
    pipe = radfoam.create_pipeline(1)
    assert pipe.gradient_row_pitch == "auto"
    f = pipe.trace_forward(p, at, adj, off, r, s)
    res = pipe.trace_backward(p, at, adj, off, r, s, f["rgba"], torch.randn_like(f["rgba"]))
    pg, ag = res["points_grad"], res["attr_grad"]
    assert not ag.is_contiguous()                      # A = 13: rows of 16 floats
    ex = rdist.SparseGradExchange()
    want_pg, want_ag = pg.clone(), ag.clone()
    send = torch.zeros((n_points, ex._pitch(13)), device="cuda:0")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ex._compact(pg, ag, send, cnt)
    k = int(cnt)
    assert 0 < k < n_points
    ex._scatter(send, k, pg, ag, zero=True)
    assert float(res["flat_grad"].abs().sum()) == 0.0
    ex._scatter(send, k, pg, ag, zero=False)
    assert torch.equal(pg + 0.0, want_pg + 0.0) and torch.equal(ag + 0.0, want_ag + 0.0)


def test_cpu_path_roundtrip():
    """not gpu: the torch-indexing path on its own (what the gloo tests exercise across ranks)."""
    n, a = 3000, 28
    ex = rdist.SparseGradExchange()
    flat, pg, ag = _case(n, a, 0.2, 1, "cpu")
    before = flat.clone()
    send = torch.zeros((n, ex._pitch(a)))
    cnt = torch.zeros(1, dtype=torch.int32)
    ex._compact(pg, ag, send, cnt)
    k = int(cnt)
    touched = ((before[: 3 * n].view(n, 3) != 0).any(1) | (before[3 * n:].view(n, a) != 0).any(1)).sum()
    assert k == int(touched)
    ex._scatter(send, k, pg, ag, zero=True)
    assert float(flat.abs().sum()) == 0.0
    ex._scatter(send, k, pg, ag, zero=False)
    assert torch.equal(flat, before)
