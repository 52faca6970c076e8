"""Input pipeline: the native BatchProvider / LIDC loader against (a) an index + annotator stream recorded from
the REAL reference BatchProvider, (b) the numpy twin of the augmentation arithmetic (oracle/augment.py - OpenCV's rules restated,
parity with cv2 itself unpinned: cv2 is absent from this image and from /root/reference)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

import unet_zoo_amd  # noqa: F401
from tests import _golden as G
from unet_zoo_amd.data import batch_provider as BP
from unet_zoo_amd.data import lidc_data_loader as LL


def _toy(N, A, H=4, W=4):
    X = np.arange(N, dtype=np.float32)[:, None, None] * np.ones((1, H, W), np.float32)
    y = np.zeros((N, H, W, A), np.uint8)
    for a in range(A):
        y[..., a] = a
    return X, y


def test_sampling_stream_equals_the_reference_batch_provider():
    """Same seed -> same batches and annotators as data/batch_provider.py:43-67,131-137 (sampling without replacement over the
    whole index set, refill when fewer than a batch remain, sorted indices, one annotator draw per sample)."""
    with open(os.path.join(G.GOLDEN, "batch_provider_stream.json")) as f:
        ref = json.load(f)
    X, y = _toy(ref["N"], ref["A"])
    np.random.seed(ref["seed"])
    bp = BP.BatchProvider(X, y, np.arange(ref["N"]), add_dummy_dimension=True, num_labels_per_subject=ref["A"],
                          annotator_range=range(ref["A"]), device="cpu")
    seen = []
    for rec in ref["batches"]:
        idx = bp._draw_indices(ref["batch"])
        ann, prm = bp._draw(idx)
        assert idx.tolist() == rec["idx"] and ann.tolist() == rec["ann"]
        assert np.array_equal(prm[:, 1], np.ones(len(idx))) and not prm[:, [0, 3, 7]].any()        # no augmentation requested
        seen += idx.tolist()
    assert sorted(set(seen[:20])) == sorted(seen[:20])                # the first four batches never repeat an image


def test_augmentation_draws_follow_the_reference_order_and_ranges():
    opts = dict(do_rotations=True, do_scaleaug=True, do_fliplr=True, do_flipud=True, nlabels=2)
    np.random.seed(3)
    prm = BP.draw_augmentation(2000, (128, 128), opts)
    # replay the reference's draw sequence (batch_provider.py:186-266) by hand with the same seed
    np.random.seed(3)
    for ii in range(2000):
        exp = np.zeros(8, np.float32); exp[1] = 1
        if np.random.randint(2) == 0:
            a = np.random.uniform(-10.0, 10.0)
            exp[0], exp[1], exp[2] = 1, np.cos(np.deg2rad(a)), np.sin(np.deg2rad(a))
            r = np.random.randint(128 - 30, 129); px = np.random.randint(0, 128 - r + 1); py = np.random.randint(0, 128 - r + 1)
            exp[3:7] = (1, px, py, r)
        f = 0
        if np.random.randint(2) == 0: f |= 1
        if np.random.randint(2) == 0: f |= 2
        exp[7] = f
        assert np.allclose(prm[ii], exp), ii
    assert 0.4 < prm[:, 0].mean() < 0.6 and prm[prm[:, 3] > 0, 6].min() >= 98 and prm[:, 6].max() <= 128
    # the experiment files' key spelling (do_flip_lr / do_flip_ud) never enables flips - reference quirk kept
    np.random.seed(0)
    assert not BP.draw_augmentation(200, (128, 128), dict(do_flip_lr=True, do_flip_ud=True, nlabels=2))[:, 7].any()
    with pytest.raises(AssertionError):
        BP.draw_augmentation(1, (8, 8), dict(do_rotations=True))            # nlabels required (batch_provider.py:176-180)


def test_lidc_preparation_from_a_pickle(tmp_path):
    rs = np.random.default_rng(0)
    data = {}
    for s in range(40):
        for k in range(rs.integers(1, 4)):
            data[f"s{s}_{k}"] = dict(image=rs.random((16, 16)).astype(np.float32), masks=[(rs.random((16, 16)) > 0.5) for _ in range(4)],
                                     series_uid=f"uid{s}")
    src = tmp_path / "lidc.pickle"
    with open(src, "wb") as f:
        pickle.dump(data, f)
    np.random.seed(0)
    d = LL.load_and_maybe_process_data(str(src), str(tmp_path / "pre"))
    n = {tt: d[tt]["images"].shape[0] for tt in d}
    assert sum(n.values()) == len(data) and all(v > 0 for v in n.values())
    assert d["train"]["labels"].shape[1:] == (16, 16, 4) and d["train"]["labels"].dtype == np.uint8
    assert -0.5 <= d["train"]["images"].min() and d["train"]["images"].max() <= 0.5                      # image - 0.5 (:92)
    subj = {tt: set(d[tt]["uids"].tolist()) for tt in d}
    assert not (subj["train"] & subj["test"]) and not (subj["train"] & subj["val"]) and not (subj["val"] & subj["test"])   # split BY subject
    assert len(subj["test"]) == 8 and len(subj["val"]) == 7                                                # 20 % of 40, then 20 % of 32
    d2 = LL.load_and_maybe_process_data(str(src), str(tmp_path / "pre"))                                   # second call: already prepared
    assert d2["test"]["images"].shape == d["test"]["images"].shape
    assert LL.crop_or_pad_slice_to_size(np.ones((6, 10)), 8, 8).shape == (8, 8)
    assert LL.crop_or_pad_slice_to_size(np.ones((6, 10)), 8, 8).sum() == 6 * 8


def test_provider_refuses_to_assemble_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check")
    X, y = _toy(8, 2)
    bp = BP.BatchProvider(X, y, np.arange(8), num_labels_per_subject=2, device="cpu")
    with pytest.raises(Exception):
        bp.next_batch(4)


@pytest.mark.gpu
def test_device_batch_assembly_matches_the_numpy_twin():
    from oracle import augment as OA
    rs = np.random.default_rng(5)
    N, H, W, A, B = 40, 128, 128, 4, 24
    yy, xx = np.mgrid[0:H, 0:W]
    X = rs.standard_normal((N, H, W)).astype(np.float32) * 0.2
    y = np.zeros((N, H, W, A), np.uint8)
    for i in range(N):
        for a in range(A):
            cy, cx, r = rs.uniform(30, 98, 2).tolist() + [rs.uniform(8, 30)]
            y[i, ..., a] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
    opts = dict(do_rotations=True, do_scaleaug=True, do_fliplr=True, do_flipud=True, nlabels=2, augment_every_nth=1)
    bp = BP.BatchProvider(X, y, np.arange(N), add_dummy_dimension=True, do_augmentations=True, augmentation_options=opts,
                          num_labels_per_subject=A, annotator_range=range(A))
    np.random.seed(11)
    idx = bp._draw_indices(B)
    ann, prm = bp._draw(idx)
    xd, sd = bp._assemble(idx, ann, prm)
    assert xd.shape == (B, 1, H, W) and sd.shape == (B, H, W)
    worst, flips_seen, diff = 0.0, set(), 0
    for b in range(B):
        ri, rl = OA.augment(X[idx[b]], y[idx[b], ..., ann[b]], prm[b], 2)
        worst = max(worst, float(np.abs(xd[b, 0].cpu().numpy() - ri).max()))
        diff += int((sd[b].cpu().numpy().astype(np.int64) != rl).sum())
        flips_seen.add(int(prm[b, 7]))
    assert worst <= 2e-5, worst
    assert diff <= 1e-4 * B * H * W, diff                      # exact up to interpolation ties at 0.5
    assert len(flips_seen) >= 3 and prm[:, 0].all() and prm[:, 3].all()
    # without augmentation the provider is a pure gather + annotator selection; numpy output on request
    plain = BP.BatchProvider(X, y, np.arange(N), add_dummy_dimension=True, num_labels_per_subject=A, annotator_range=range(A))
    np.random.seed(2)
    xb, sb = plain.next_batch(8, host=True)
    np.random.seed(2)
    replay = BP.BatchProvider(X, y, np.arange(N), add_dummy_dimension=True, num_labels_per_subject=A, annotator_range=range(A))
    i2 = replay._draw_indices(8); a2, _ = replay._draw(i2)
    assert np.array_equal(xb[:, 0], X[i2]) and all(np.array_equal(sb[k], y[i2[k], ..., a2[k]]) for k in range(8))


def _prep_worker(rank, world, port, src, pre, out_dir):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(1234 + rank)                    # an unseeded split would now differ per rank
    d = LL.load_and_maybe_process_data(src, pre)
    np.save(os.path.join(out_dir, f"uids{rank}.npy"), np.concatenate([np.sort(np.unique(d[tt]["uids"])) for tt in ("train", "val", "test")]))
    assert isinstance(d["train"]["images"], np.memmap)             # really memory-mapped: one host copy per process at most
    dist.barrier()
    dist.destroy_process_group()


def test_lidc_preparation_under_two_ranks_is_done_once_and_agrees(tmp_path):
    """every rank used to prepare at once (unseeded split, racing writes).  Now rank 0 prepares into temporary
    names, the others wait at a barrier, and the split is seeded."""
    import socket
    import torch.multiprocessing as mp
    rs = np.random.default_rng(1)
    data = {f"s{s}_{k}": dict(image=rs.random((8, 8)).astype(np.float32), masks=[(rs.random((8, 8)) > 0.5) for _ in range(4)], series_uid=f"uid{s}")
            for s in range(30) for k in range(2)}
    src = tmp_path / "lidc.pickle"
    with open(src, "wb") as f:
        pickle.dump(data, f)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_prep_worker, args=(2, port, str(src), str(tmp_path / "pre"), str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "uids0.npy"), np.load(tmp_path / "uids1.npy"))
    assert not [p for p in os.listdir(tmp_path / "pre") if ".tmp" in p]                         # temporaries renamed into place
    assert sorted(os.listdir(tmp_path / "pre")) == sorted(f"data_lidc_{tt}_{a}.npy" for tt in ("train", "test", "val") for a in LL.ARRAYS)
