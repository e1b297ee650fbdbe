"""-m gpu: SURVEY f2 on the device.  `csr_permute_symmetric` and `reorder_and_save` run their heavy steps
(vertex order, symmetric CSR relabel + coalesce, split bucketing, feature-row gather through the HIP
kernel) on the GPU here and are compared with a scipy / numpy restatement of driver/dataset.py:289-297
(relabel + coalesce) and :299-353 (ordering rule, per-partition files)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _graph(n, m, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    src = (n * rng.random(m) ** 2).astype(np.int64)           # skewed: a few hubs
    dst = rng.integers(0, n, size=m)
    a = sp.coo_matrix((np.ones(2 * m), (np.r_[src, dst], np.r_[dst, src])), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a, rng


def test_csr_permute_symmetric_on_gpu_matches_restatement():
    import scipy.sparse as sp
    from salient_plusplus_amd.dataset import csr_permute_symmetric
    n = 30_000
    a, rng = _graph(n, 300_000, 5)
    perm = rng.permutation(n)
    invperm = np.argsort(perm)
    rp, cl = csr_permute_symmetric(torch.from_numpy(a.indptr.astype(np.int64)).cuda(),
                                   torch.from_numpy(a.indices.astype(np.int64)).cuda(), torch.from_numpy(invperm).cuda())
    assert rp.is_cuda and cl.is_cuda
    coo = a.tocoo()
    b = sp.coo_matrix((np.ones(coo.nnz), (invperm[coo.row], invperm[coo.col])), shape=(n, n)).tocsr()
    b.sum_duplicates()
    b.sort_indices()
    np.testing.assert_array_equal(rp.cpu().numpy(), b.indptr)
    np.testing.assert_array_equal(cl.cpu().numpy(), b.indices)
    # with duplicated input entries the result is coalesced (dataset.py:296 .coalesce())
    rp2, cl2 = csr_permute_symmetric(torch.tensor([0, 3, 4, 6]).cuda(), torch.tensor([1, 1, 2, 0, 0, 0]).cuda(),
                                     torch.tensor([2, 0, 1]).cuda())
    assert rp2.tolist() == [0, 1, 2, 4] and cl2.tolist() == [2, 2, 0, 1]


@pytest.mark.parametrize("prob_kind", ["1d", "2d", "none"])
def test_reorder_and_save_on_gpu_matches_restatement(tmp_path, prob_kind):
    import scipy.sparse as sp
    from salient_plusplus_amd.dataset import DisjointPartFeatReorderedDataset as D
    from salient_plusplus_amd.dataset import FastDataset
    n, P, F = 20_000, 4, 100
    a, rng = _graph(n, 150_000, 11)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, size=(n, 1)))
    order = rng.permutation(n)
    split = {"train": torch.from_numpy(order[:9000]), "valid": torch.from_numpy(order[9000:12000]),
             "test": torch.from_numpy(order[12000:])}
    ds = FastDataset.from_tensors("toy", x, y, torch.from_numpy(a.indptr.astype(np.int64)),
                                  torch.from_numpy(a.indices.astype(np.int64)), split, 9)
    labels = torch.from_numpy(rng.integers(0, P, size=n))
    if prob_kind == "1d":
        prob = torch.from_numpy(rng.permutation(n).astype(np.float64) / (2 * n))          # distinct, in [0, .5)
        eff = prob.numpy()
    elif prob_kind == "2d":
        prob = torch.from_numpy(np.stack([rng.permutation(n).astype(np.float64) / (2 * n) for _ in range(P)]))
        eff = prob.numpy()[labels.numpy(), np.arange(n)]                                  # row p for partition p's vertices
    else:
        prob, eff = None, np.zeros(n)
    out = D.reorder_and_save(ds, labels, prob, tmp_path, device=torch.device("cuda", 0))
    assert out == tmp_path / f"metis-reordered-k{P}" / "toy"
    assert sorted(p.name for p in out.iterdir()) == sorted(
        [f + ".pt" for f in ("num_parts", "rowptr", "col", "split_idx", "split_idx_parts", "part_offsets", "y",
                             "meta_info", "name")] + [f"x{r}.pt" for r in range(P)])
    # --- restatement (the reference's single float key, :309-323; stable so that ties are comparable) ---
    key = 2.0 * (labels.numpy().max() - labels.numpy()) + eff
    perm = np.argsort(-key, kind="stable")
    invperm = np.argsort(perm, kind="stable")
    coo = a.tocoo()
    b = sp.coo_matrix((np.ones(coo.nnz), (invperm[coo.row], invperm[coo.col])), shape=(n, n)).tocsr()
    b.sum_duplicates()
    b.sort_indices()
    sizes = np.bincount(labels.numpy(), minlength=P)
    for r in range(P):
        got = D.from_path(tmp_path / f"metis-reordered-k{P}", "toy", r)
        np.testing.assert_array_equal(got.rowptr.numpy(), b.indptr)
        np.testing.assert_array_equal(got.col.numpy(), b.indices)
        np.testing.assert_array_equal(got.part_offsets.numpy(), np.r_[0, np.cumsum(sizes)])
        lo, hi = int(got.part_offsets[r]), int(got.part_offsets[r + 1])
        np.testing.assert_array_equal(got.x.numpy().view(np.uint16), ds.x.numpy()[perm][lo:hi].view(np.uint16))
        np.testing.assert_array_equal(got.y.numpy(), ds.y.numpy()[perm])
        assert got.split_idx == {}
        for k in ("train", "valid", "test"):
            ids = got.split_idx_parts[r][k].numpy()
            want = invperm[ds.split_idx[k].numpy()]
            np.testing.assert_array_equal(np.sort(ids), np.sort(want[(want >= lo) & (want < hi)]))
