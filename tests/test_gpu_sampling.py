"""The label-only half of the InfoNCE loss on the device (csrc/sampling.hip; reference utils/loss_functions.py:484-552): cell validity and
matches against the PyTorch formulation of the same steps (warp_image_batch / getMasks / warp_points, which restate the reference lines),
the uniform draw of the matched cells, the negatives with the reference's redraw rule, and the counting sort into CSR against
torch.sort(stable) + searchsorted.  Integer / index work: exact, except where a pixel coordinate sits within an fp32 rounding of a
.5 boundary (the two formulations order their fp32 operations differently): <= 0.1 % of the cells may differ under a general homography,
none under a translation by whole pixels."""
import pytest
import torch

from yolopoint_amd import _hip
from yolopoint_amd.utils import loss_functions as LF
from yolopoint_amd.utils.utils import getMasks

pytestmark = pytest.mark.gpu


def _cells(mask, inv_h):
    B, _, H, W = mask.shape
    Nc = (H // 8) * (W // 8)
    valid = torch.empty(B * Nc, dtype=torch.uint8, device=mask.device)
    uvb = torch.empty(B * Nc, 2, dtype=torch.float32, device=mask.device)
    _hip.check(_hip.lib().yp_nce_cells(mask.data_ptr(), inv_h.data_ptr(), B, H, W, valid.data_ptr(), uvb.data_ptr(), _hip.stream_ptr()))
    return valid.view(B, Nc), uvb.view(B, Nc, 2)


def _torch_cells(mask, inv_h):
    B, _, H, W = mask.shape
    Hc, Wc = H // 8, W // 8
    dev = mask.device
    valid = LF.warp_image_batch(mask, inv_h, mode='nearest', device=dev)
    valid = (getMasks(valid, dev, 8) == 1.).flatten(1, -1)
    uv_a = LF.get_coor_cells(Hc, Wc, uv=True).to(dev)
    uv_b = LF.warp_points(uv_a, LF.homography_scaling(inv_h, Hc, Wc, device=dev), dev).round_()
    return valid, uv_b


@pytest.mark.parametrize("B,H,W,kind", [(3, 128, 192, "identity"), (2, 256, 256, "shift"), (4, 128, 128, "general"), (2, 640, 640, "general")])
def test_cell_validity_and_matches(cuda, B, H, W, kind):
    g = torch.Generator(device=cuda).manual_seed(H + B)
    mask = torch.ones(B, 1, H, W, device=cuda)
    mask[:, :, H // 3:H // 3 + 20, W // 4:W // 4 + 50] = 0.0
    mask[:, :, :, -9:] = 0.0
    inv_h = torch.eye(3, device=cuda).repeat(B, 1, 1)
    if kind == "shift":       # 16 pixels in x, -8 in y, in normalised units of an align_corners grid
        inv_h[:, 0, 2] = 2.0 * 16 / (W - 1)
        inv_h[:, 1, 2] = -2.0 * 8 / (H - 1)
    elif kind == "general":
        inv_h = inv_h + (torch.rand(B, 3, 3, device=cuda, generator=g) - 0.5) * torch.tensor([[0.2, 0.2, 0.3], [0.2, 0.2, 0.3], [0.05, 0.05, 0.0]], device=cuda)
    valid, uvb = _cells(mask, inv_h.contiguous())
    ref_valid, ref_uvb = _torch_cells(mask, inv_h)
    bad_v = int((valid.bool() != ref_valid).sum())
    bad_u = int((uvb != ref_uvb).any(-1).sum())
    total = valid.numel()
    assert 0 < int(valid.sum()) < total
    if kind == "general":
        assert bad_v <= 1e-3 * total and bad_u <= 1e-3 * total, (bad_v, bad_u, total)
    else:
        assert bad_v == 0 and (bad_u == 0 or kind == "shift"), (bad_v, bad_u)
        assert bad_u <= 1e-3 * total


def _fixture_cases():
    """The inputs of tests/golden/nce_cells.npz, regenerated exactly as tests/golden/make_golden.py::nce_cells_cases builds them."""
    cases = []
    for name, B, H, W, kind in (("identity", 3, 128, 192, "identity"), ("shift", 2, 256, 256, "shift"), ("general", 4, 128, 128, "general"),
                                ("general_320", 2, 320, 320, "general")):
        g = torch.Generator().manual_seed(H + B)
        mask = torch.ones(B, 1, H, W)
        mask[:, :, H // 3:H // 3 + 20, W // 4:W // 4 + 50] = 0.0
        mask[:, :, :, -9:] = 0.0
        inv_h = torch.eye(3).repeat(B, 1, 1)
        if kind == "shift":
            inv_h[:, 0, 2] = 2.0 * 16 / (W - 1)
            inv_h[:, 1, 2] = -2.0 * 8 / (H - 1)
        elif kind == "general":
            inv_h = inv_h + (torch.rand(B, 3, 3, generator=g) - 0.5) * torch.tensor([[0.2, 0.2, 0.3], [0.2, 0.2, 0.3], [0.05, 0.05, 0.0]])
        cases.append((name, kind, mask, inv_h.contiguous()))
    return cases


@pytest.mark.parametrize("idx", range(4))
def test_cell_validity_and_matches_against_the_reference_fixture(cuda, idx):
    """yp_nce_cells against tests/golden/nce_cells.npz: the valid-cell mask and the rounded warped cell coordinates produced by the
    REFERENCE's own functions (warp_image_batch / getMasks / homography_scaling / warp_points, utils/loss_functions.py:499-523, imported in
    tests/golden/make_golden.py::gen_nce_cells).  Exact for the identity and for a translation by whole pixels; under a general homography a
    cell whose pixel coordinate sits within one fp32 rounding of a .5 boundary may land on the other side: <= 0.1 % of the cells."""
    import os
    import numpy as np
    name, kind, mask, inv_h = _fixture_cases()[idx]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nce_cells.npz"))
    assert np.array_equal(gold[name + ".inv_h"], inv_h.numpy())           # same inputs as the generator's
    B, _, H, W = mask.shape
    Nc = (H // 8) * (W // 8)
    ref_valid = torch.from_numpy(np.unpackbits(gold[name + ".valid"], axis=-1)[:, :Nc].astype(np.bool_))
    ref_uvb = torch.from_numpy(gold[name + ".uv_b"].astype(np.float32))
    valid, uvb = _cells(mask.to(cuda), inv_h.to(cuda))
    valid, uvb = valid.bool().cpu(), uvb.cpu()
    bad_v = int((valid != ref_valid).sum())
    # coordinates matter where the reference keeps the cell (the others are dropped by the mask)
    bad_u = int(((uvb != ref_uvb).any(-1) & ref_valid).sum())
    total = valid.numel()
    if kind == "general":
        assert bad_v <= 1e-3 * total and bad_u <= 1e-3 * total, (name, bad_v, bad_u, total)
    else:
        assert bad_v == 0 and bad_u == 0, (name, bad_v, bad_u)


def test_uniform_draw_of_the_matched_cells(cuda):
    B, Hc, Wc, samples = 3, 12, 16, 40
    Nc = Hc * Wc
    g = torch.Generator(device=cuda).manual_seed(3)
    valid = (torch.rand(B, Nc, device=cuda, generator=g) < 0.6).to(torch.uint8)
    valid[1, 50:] = 0                                         # image 1: 50 candidates at most
    valid[1, :50] = 1
    valid[1, 7] = 0                                           # -> 49 valid cells
    uvb = torch.randint(0, 16, (B, Nc, 2), device=cuda, generator=g).float()
    counts = torch.zeros(B, Nc, device=cuda)
    for big, expect_pool in ((40, 40), (64, 49)):
        for seed in range(300 if big == 40 else 3):
            store = torch.full((2 * B * big * 2,), -7.0, device=cuda)
            meta = torch.empty(4, dtype=torch.int32, device=cuda)
            _hip.check(_hip.lib().yp_nce_select(valid.data_ptr(), uvb.data_ptr(), B, Hc, Wc, big, 1000 + seed, store.data_ptr(), meta.data_ptr(), _hip.stream_ptr()))
            pool, n = int(meta[0]), int(meta[1])
            assert pool == expect_pool and n == B * pool and int(meta[2]) == 0
            uab = store[:2 * n * 2].view(2 * B, pool, 2)
            tail = store[2 * n * 2:]
            assert tail.numel() == 0 or float((tail + 7.0).abs().max()) == 0.0               # nothing written behind the compact list
            cx = torch.round((uab[:B, :, 0] + 1) / 2 * Wc).long()
            cy = torch.round((uab[:B, :, 1] + 1) / 2 * Hc).long()
            cell = cy * Wc + cx
            assert bool((cell[:, 1:] > cell[:, :-1]).all())                                  # distinct, in cell order
            assert bool(torch.gather(valid.long(), 1, cell).all())                           # only valid cells
            want_b = torch.gather(uvb, 1, cell.unsqueeze(-1).expand(-1, -1, 2))
            got_b = torch.stack(((uab[B:, :, 0] + 1) / 2 * Wc, (uab[B:, :, 1] + 1) / 2 * Hc), -1)
            assert float((got_b - want_b).abs().max()) < 1e-4
            if big == 40:
                counts.scatter_add_(1, cell, torch.ones_like(cell, dtype=torch.float32))
    # uniform without replacement: every valid cell of image b is drawn with probability 40 / #valid_b
    for b in range(B):
        nv = int(valid[b].sum())
        p = 40.0 / nv
        freq = counts[b][valid[b].bool()] / 300.0
        assert float(counts[b][~valid[b].bool()].sum()) == 0.0
        sigma = (p * (1 - p) / 300.0) ** 0.5
        assert float((freq - p).abs().max()) < 5.0 * sigma + 1e-6, (b, float((freq - p).abs().max()), sigma)


def test_negatives_follow_the_reference_rule(cuda):
    n, negs = 5000, 60
    meta = torch.zeros(4, dtype=torch.int32, device=cuda)
    idx = torch.full((n, negs + 1), -1, dtype=torch.int32, device=cuda)
    _hip.check(_hip.lib().yp_nce_negatives(n, negs, 77, meta.data_ptr(), idx.data_ptr(), 0, _hip.stream_ptr()))
    idx2 = torch.empty_like(idx)
    meta2 = torch.zeros(4, dtype=torch.int32, device=cuda)
    _hip.check(_hip.lib().yp_nce_negatives(n, negs, 77, meta2.data_ptr(), idx2.data_ptr(), 0, _hip.stream_ptr()))
    assert torch.equal(idx, idx2) and int(meta[2]) == int(meta2[2])
    assert torch.equal(idx[:, 0].long(), torch.arange(n, device=cuda))
    r = idx[:, 1:].long()
    assert int(r.min()) >= 0 and int(r.max()) < n
    same = int(meta[2])                                       # draws that hit their own row: n * negs / n = negs expected
    assert 20 <= same <= 120
    # those were replaced by values in [0, same): rows >= same keep no self-reference at all
    own = r == torch.arange(n, device=cuda).unsqueeze(1)
    assert int(own[same:].sum()) == 0
    # uniform over [0, n): mean and a coarse histogram
    assert abs(float(r.float().mean()) - (n - 1) / 2) < 0.01 * n
    hist = torch.bincount(r.flatten() * 10 // n, minlength=10).float()
    assert float((hist / hist.sum() - 0.1).abs().max()) < 0.005
    _hip.check(_hip.lib().yp_nce_negatives(n, negs, 78, meta2.zero_().data_ptr(), idx2.data_ptr(), 0, _hip.stream_ptr()))
    assert not torch.equal(idx, idx2)


@pytest.mark.parametrize("n_items,n_buckets,wide", [(100000, 500, True), (4096, 20000, False), (37, 5, True), (60000, 48000, False), (9000, 3, True), (40000, 100, True), (27000, 100, True), (300000, 24000, True)])
def test_counting_sort_into_csr(cuda, n_items, n_buckets, wide):
    g = torch.Generator(device=cuda).manual_seed(n_items)
    keys = torch.randint(0, n_buckets, (n_items,), device=cuda, generator=g).to(torch.int32)
    keys[torch.rand(n_items, device=cuda, generator=g) < 0.05] = 2 ** 31 - 1          # skipped entries (taps outside the map)
    order, offsets = LF._csr(keys, n_buckets, wide)
    skeys, ref_order = torch.sort(keys, stable=True)
    ref_off = torch.searchsorted(skeys, torch.arange(n_buckets + 1, device=cuda, dtype=torch.int32))
    assert torch.equal(offsets.long(), ref_off)
    kept = int(ref_off[-1])
    assert torch.equal(order[:kept].long(), ref_order[:kept])


def test_prepare_is_reproducible_and_feeds_the_loss(cuda):
    """infonce_prepare end to end: same torch seed -> same structures; the loss computed from them equals the PyTorch Gram-matrix
    formulation on the same structures."""
    B, Hc, Wc, D = 2, 16, 16, 64
    mask = torch.ones(B, 1, Hc * 8, Wc * 8, device=cuda)
    mask[:, :, 40:70, 30:90] = 0.0
    Hinv = torch.eye(3, device=cuda).repeat(B, 1, 1)
    Hinv[:, 0, 2] = 0.1
    torch.manual_seed(5)
    a = LF.infonce_prepare(mask, Hinv, (B, D, Hc, Wc), True, 60, 20, 8, cuda, pair_index=True)
    torch.manual_seed(5)
    b = LF.infonce_prepare(mask, Hinv, (B, D, Hc, Wc), True, 60, 20, 8, cuda, pair_index=True)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    for (i1, o1, f1), (i2, o2, f2) in ((a[3], b[3]), (a[4], b[4])):      # (entries behind offsets[-1] -- skipped taps -- are not written)
        assert torch.equal(i1, i2) and torch.equal(f1, f2) and torch.equal(o1[:int(f1[-1])], o2[:int(f2[-1])])
    c = LF.infonce_prepare(mask, Hinv, (B, D, Hc, Wc), True, 60, 20, 8, cuda, pair_index=True)
    assert not torch.equal(a[0], c[0])
    desc = torch.nn.functional.normalize(torch.randn(2 * B, Hc, Wc, D, device=cuda), dim=-1).permute(0, 3, 1, 2)
    got = LF.infonce(desc[:B], desc[B:], mask, Hinv, device=cuda, prepared=a, descriptors_pair=desc, num_samples_per_image=60, num_masked_non_matches_per_match=20)
    ua, ub, rnd = a[:3]
    da = torch.nn.functional.grid_sample(desc[:B], ua.unsqueeze(1), mode='bilinear', align_corners=True).squeeze(2).transpose(1, 2).flatten(0, 1)
    db = torch.nn.functional.grid_sample(desc[B:], ub.unsqueeze(1), mode='bilinear', align_corners=True).squeeze(2).transpose(1, 2).flatten(0, 1)
    logits = torch.cat(((da * db).sum(-1, keepdim=True), (da @ db.t()).gather(1, rnd.long())), 1) / 0.07
    ref = -torch.log_softmax(logits, 1)[:, 0].mean()
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))


@pytest.mark.parametrize("B,Hc,Wc,samples,negs", [(1, 8, 20, 50, 7), (5, 160, 160, 3000, 40), (2, 24, 16, 300, 200)])
def test_prepare_shapes_and_scarce_cells(cuda, B, Hc, Wc, samples, negs):
    """One image, rectangular maps, a 1280 x 1280 image (25 600 cells per image in the selection kernel's LDS) and masks that leave fewer valid
    cells than `samples`: pool = min(samples, fewest valid cells); every index structure is consistent with it."""
    mask = torch.ones(B, 1, Hc * 8, Wc * 8, device=cuda)
    mask[-1, :, : (Hc * 8) // 2 + 8] = 0.0                      # the last image keeps less than half of its cells
    Hinv = torch.eye(3, device=cuda).repeat(B, 1, 1)
    torch.manual_seed(B + Hc)
    ua, ub, rnd, (idx, order, offsets), (uab, s_order, s_offsets) = LF.infonce_prepare(mask, Hinv, (B, 64, Hc, Wc), True, samples, negs, 8, cuda, pair_index=True)
    valid_last = (Hc - (Hc // 2 + 1)) * Wc
    pool = min(samples, valid_last)
    n = B * pool
    assert tuple(ua.shape) == (B, pool, 2) and tuple(ub.shape) == (B, pool, 2) and tuple(uab.shape) == (2 * B, pool, 2)
    assert tuple(idx.shape) == (n, negs + 1) and tuple(rnd.shape) == (n, negs) and offsets.numel() == n + 1 and int(offsets[-1]) == n * (negs + 1)
    assert torch.equal(ua, ub)                                  # identity homography: every cell matches itself
    cy = torch.round((ua[..., 1] + 1) / 2 * Hc).long()
    assert int(cy[-1].min()) >= Hc // 2 + 1                     # only valid cells of the masked image
    # the transposed edge list really is the transpose
    flat = idx.flatten().long()
    assert torch.equal(flat[order.long()], torch.repeat_interleave(torch.arange(n, device=cuda), (offsets[1:] - offsets[:-1]).long()))
    # the tap list: 2 * n points, up to four bilinear taps each (normPts divides by the map size, grid_sample's align_corners maps [-1, 1] to
    # [0, size - 1]: cell x is sampled at x * (W - 1) / W, as in the reference), grouped by cell
    assert 2 * n <= int(s_offsets[-1]) <= 8 * n and s_offsets.numel() == 2 * B * Hc * Wc + 1
    taps = s_order[:int(s_offsets[-1])].long()
    assert int(taps.min()) >= 0 and int(taps.max()) < 8 * n and taps.unique().numel() == taps.numel()


@pytest.mark.parametrize("scarce", [False, True])
def test_prepare_without_host_synchronisation_equals_the_synchronising_form(cuda, scarce):
    """infonce_prepare(sync=False): counts stay on the device (meta[0] = points per image, meta[1] = matched rows), arrays keep their capacity;
    their valid prefixes equal what the synchronising form returns for the same seed, rows behind the counts carry keys the sort skips."""
    B, Hc, Wc, D, samples, negs = 3, 16, 20, 64, 120, 25
    mask = torch.ones(B, 1, Hc * 8, Wc * 8, device=cuda)
    if scarce:
        mask[1, :, 40:] = 0.0                                   # image 1 keeps 5 cell rows: 100 valid cells < samples
    Hinv = torch.eye(3, device=cuda).repeat(B, 1, 1)
    Hinv[:, 1, 2] = -0.05
    torch.manual_seed(8)
    ua, ub, rnd, (idx, order, offsets), (uab, s_order, s_offsets) = LF.infonce_prepare(mask, Hinv, (B, D, Hc, Wc), True, samples, negs, 8, cuda, pair_index=True)
    torch.manual_seed(8)
    d = LF.infonce_prepare(mask, Hinv, (B, D, Hc, Wc), True, samples, negs, 8, cuda, pair_index=True, sync=False)
    pool, n = int(d["meta"][0]), int(d["meta"][1])
    assert pool == ua.shape[1] and n == B * pool and (pool < samples) == scarce
    E = negs + 1
    assert torch.equal(d["uab"][:4 * n].view(2 * B, pool, 2), uab)
    assert torch.equal(d["idx"][:n], idx) and bool((d["idx"][n:] == 2 ** 31 - 1).all())
    assert torch.equal(d["offsets"][:n + 1], offsets) and bool((d["offsets"][n:] == n * E).all())
    assert torch.equal(d["order"][:n * E], order)
    assert torch.equal(d["s_offsets"], s_offsets) and torch.equal(d["s_order"][:int(s_offsets[-1])], s_order[:int(s_offsets[-1])])


@pytest.fixture
def capped_grids():
    """yp_sampling_set_max_workgroups(3): the large launches run as 3 workgroups that walk their items (what a training step does with 256,
    so that label work beside the forward leaves the CU slots to the convolutions)."""
    from yolopoint_amd import _hip
    _hip.check(_hip.lib().yp_sampling_set_max_workgroups(3))
    yield
    _hip.check(_hip.lib().yp_sampling_set_max_workgroups(0))


def test_capped_grids_give_the_same_results(cuda, capped_grids):
    for args in [(100000, 500, True), (60000, 48000, False), (9000, 3, True)]:
        test_counting_sort_into_csr(cuda, *args)
    test_cell_validity_and_matches(cuda, 2, 256, 256, "general")
    test_prepare_without_host_synchronisation_equals_the_synchronising_form(cuda, True)
    test_prepare_is_reproducible_and_feeds_the_loss(cuda)
