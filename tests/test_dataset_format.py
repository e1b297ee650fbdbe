"""CPU: on-disk dataset formats (driver/dataset.py:29-142, :145-427) -- save/load round
trips in the reference's directory layout and the vertex reordering against a scipy restatement of
the nested ``csr_permute_symmetric`` (dataset.py:289-297) and of the ordering rule (:299-323)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _dataset(n=400, seed=3):
    from salient_plusplus_amd.dataset import FastDataset
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=3000)
    dst = rng.integers(0, n, size=3000)
    import scipy.sparse as sp
    a = sp.coo_matrix((np.ones(6000), (np.r_[src, dst], np.r_[dst, src])), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    x = torch.from_numpy(rng.standard_normal((n, 12)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 7, size=(n, 1)))
    perm = rng.permutation(n)
    split = {"train": torch.from_numpy(perm[:200]), "valid": torch.from_numpy(perm[200:300]),
             "test": torch.from_numpy(perm[300:])}
    return FastDataset.from_tensors("toy", x, y, torch.from_numpy(a.indptr.astype(np.int64)),
                                    torch.from_numpy(a.indices.astype(np.int64)), split, 7), a


def test_fast_dataset_round_trip(tmp_path):
    from salient_plusplus_amd.dataset import FastDataset
    ds, _ = _dataset()
    assert ds.x.dtype == torch.float16 and ds.y.dim() == 1
    ds.save(tmp_path)
    assert sorted(p.name for p in (tmp_path / "toy").iterdir()) == sorted(f + ".pt" for f in FastDataset._fields)
    back = FastDataset.from_path(tmp_path, "toy")
    for f in ("x", "y", "rowptr", "col"):
        assert torch.equal(getattr(back, f), getattr(ds, f))
    assert back.num_classes == 7 and back.num_features == 12 and back.num_nodes == 400
    assert back.get_num_iterations(64) == {"train": 3, "valid": 1, "test": 1}
    nofeat = FastDataset.from_path(tmp_path, "toy", skip_features=True)
    assert nofeat.x.numel() == 0 and torch.equal(nofeat.col, ds.col)
    rp, cl, _ = back.adj_t().csr()
    assert torch.equal(rp, ds.rowptr) and torch.equal(cl, ds.col)


def test_reorder_and_save_matches_restatement(tmp_path):
    import scipy.sparse as sp
    from salient_plusplus_amd.dataset import DisjointPartFeatReorderedDataset as D
    ds, a = _dataset()
    n, P = ds.num_nodes, 3
    rng = np.random.default_rng(9)
    labels = torch.from_numpy(rng.integers(0, P, size=n))
    prob = torch.from_numpy(rng.permutation(n).astype(np.float64) / (2 * n))     # distinct values in [0, .5)
    out = D.reorder_and_save(ds, labels, prob, tmp_path, device="cpu")
    assert out == tmp_path / "metis-reordered-k3" / "toy"
    # --- restatement: ordering (dataset.py:309-323), symmetric relabel + coalesce (:289-297) ---
    ordering = (2 * (labels.max() - labels.float()) + prob.float()).numpy()
    perm = np.argsort(-ordering, kind="stable")
    invperm = np.argsort(perm, kind="stable")
    coo = a.tocoo()
    b = sp.coo_matrix((np.ones(coo.nnz), (invperm[coo.row], invperm[coo.col])), shape=(n, n)).tocsr()
    b.sum_duplicates()
    b.sort_indices()
    sizes = np.bincount(labels.numpy(), minlength=P)
    for r in range(P):
        got = D.from_path(tmp_path / "metis-reordered-k3", "toy", r)
        assert got.rank == r and got.num_parts == P
        np.testing.assert_array_equal(got.rowptr.numpy(), b.indptr)
        np.testing.assert_array_equal(got.col.numpy(), b.indices)
        np.testing.assert_array_equal(got.part_offsets.numpy(), np.r_[0, np.cumsum(sizes)])
        lo, hi = int(got.part_offsets[r]), int(got.part_offsets[r + 1])
        np.testing.assert_array_equal(got.x.numpy().view(np.uint16), ds.x.numpy()[perm][lo:hi].view(np.uint16))
        np.testing.assert_array_equal(got.y.numpy(), ds.y.numpy()[perm])
        assert got.split_idx == {}                                     # the reference saves an empty dict (:326)
        # partition r owns [lo, hi); inside it the access probability is non-increasing
        assert (labels.numpy()[perm][lo:hi] == r).all()
        assert (np.diff(prob.numpy()[perm][lo:hi]) <= 0).all()
        for k in ("train", "valid", "test"):
            ids = got.split_idx_parts[r][k].numpy()
            assert ((ids >= lo) & (ids < hi)).all()
            want = invperm[ds.split_idx[k].numpy()]
            np.testing.assert_array_equal(np.sort(ids), np.sort(want[(want >= lo) & (want < hi)]))
        pb = got.get_RangePartitionBook()
        assert pb.rank == r and pb.world_size == P
        assert got.get_num_iterations(64) == {"train": 3, "valid": 1, "test": 1}
    # relabelling preserves the graph: edge (u, v) exists iff (invperm[u], invperm[v]) exists
    assert b.nnz == a.nnz


def test_csr_permute_symmetric_coalesces_duplicates():
    from salient_plusplus_amd.dataset import csr_permute_symmetric
    rowptr = torch.tensor([0, 3, 4, 6])
    col = torch.tensor([1, 1, 2, 0, 0, 0])                 # duplicates (0,1) and (2,0)
    inv = torch.tensor([2, 0, 1])
    rp, cl = csr_permute_symmetric(rowptr, col, inv)
    # old edges {(0,1),(0,2),(1,0),(2,0)} -> new {(2,0),(2,1),(0,2),(1,2)}
    assert rp.tolist() == [0, 1, 2, 4] and cl.tolist() == [2, 2, 0, 1]


def test_synthetic_graph_is_independent_of_the_sort_range_count(monkeypatch):
    """papers-scale graphs are coalesced per row range (no sort may see 2^31 keys); the result must be
    the graph a single global sort gives."""
    from salient_plusplus_amd import synthetic
    one = synthetic.make_graph(6000, 50000, 11, torch.device("cpu"))
    monkeypatch.setattr(synthetic, "MAX_KEYS_PER_SORT", 1 << 13)
    many = synthetic.make_graph(6000, 50000, 11, torch.device("cpu"))
    assert torch.equal(one[0], many[0]) and torch.equal(one[1], many[1])
    rp, col = one
    assert int(rp[-1]) == col.numel() and bool((rp[1:] >= rp[:-1]).all())
    # symmetric and coalesced
    n = rp.numel() - 1
    row = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    key = row * n + col
    assert torch.equal(key, torch.unique(key)) and torch.equal(torch.sort(col * n + row).values, key)
