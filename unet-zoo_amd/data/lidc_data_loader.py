"""LIDC-IDRI preparation - counterpart of the reference's ``data/lidc_data_loader.py``: the pickled crops
(``{key: {'image', 'masks', 'series_uid', ...}}``) are split by subject (80 / 20 test, then 80 / 20 validation, sklearn's
``train_test_split`` exactly as :72-73) and written once as arrays ``images`` (float, image - 0.5 :92), ``labels`` (uint8, H x W x 4
annotators :94-97) and ``uids`` per split.  The reference stores them as HDF5; h5py is not available in this image and the arrays
are going to live in HBM anyway, so the container is one ``.npy`` per split and array (``data_lidc_<split>_<name>.npy``, really
memory-mapped on load).  The returned mapping has the same ``data[tt]['images' | 'labels' | 'uids']`` surface the reference's
callers index (lidc_data.py:22-53).

Two deliberate differences from the reference, both for one-process-per-GPU runs: the subject split is SEEDED (``split_seed``,
default 0; the reference's unseeded split would differ per rank and per rerun, mixing train and test subjects across ranks), and
only rank 0 prepares - into temporary names that are renamed into place - while the other ranks wait at a barrier."""
import os
import pickle

import numpy as np


def crop_or_pad_slice_to_size(slice, nx, ny):
    """lidc_data_loader.py:16-36: centre crop / zero pad a 2-D slice to (nx, ny)."""
    x, y = slice.shape
    x_s, y_s, x_c, y_c = (x - nx) // 2, (y - ny) // 2, (nx - x) // 2, (ny - y) // 2
    if x > nx and y > ny:
        return slice[x_s:x_s + nx, y_s:y_s + ny]
    out = np.zeros((nx, ny))
    if x <= nx and y > ny:
        out[x_c:x_c + x, :] = slice[:, y_s:y_s + ny]
    elif x > nx and y <= ny:
        out[:, y_c:y_c + y] = slice[x_s:x_s + nx, :]
    else:
        out[x_c:x_c + x, y_c:y_c + y] = slice[:, :]
    return out


def find_subset_for_id(ids_dict, id):
    for tt in ("test", "train", "val"):
        if id in ids_dict[tt]:
            return tt
    raise ValueError("id was not found in any of the train/test/val subsets.")


ARRAYS = ("images", "labels", "uids")


def split_paths(folder):
    return {tt: {a: os.path.join(folder, f"data_lidc_{tt}_{a}.npy") for a in ARRAYS} for tt in ("train", "test", "val")}


def prepare_data(input_file, output_folder, split_seed=0):
    """lidc_data_loader.py:47-110."""
    from sklearn.model_selection import train_test_split
    max_bytes = 2 ** 31 - 1
    bytes_in = bytearray(0)
    input_size = os.path.getsize(input_file)
    with open(input_file, "rb") as f_in:
        for _ in range(0, input_size, max_bytes):
            bytes_in += f_in.read(max_bytes)
    data = pickle.loads(bytes_in)
    unique_subjects = np.unique([v["series_uid"] for v in data.values()])
    split_ids = {}
    train_and_val_ids, split_ids["test"] = train_test_split(unique_subjects, test_size=0.2, random_state=split_seed)
    split_ids["train"], split_ids["val"] = train_test_split(train_and_val_ids, test_size=0.2, random_state=split_seed)
    images, labels, uids = ({tt: [] for tt in ("train", "test", "val")} for _ in range(3))
    for value in data.values():
        tt = find_subset_for_id(split_ids, value["series_uid"])
        images[tt].append(value["image"].astype(float) - 0.5)
        labels[tt].append(np.asarray(value["masks"]).transpose((1, 2, 0)))       # 4 x H x W -> H x W x 4
        uids[tt].append(hash(value["series_uid"]))
    os.makedirs(output_folder, exist_ok=True)
    paths = split_paths(output_folder)
    for tt in ("test", "train", "val"):
        arrays = dict(uids=np.asarray(uids[tt], dtype=np.int64), labels=np.asarray(labels[tt], dtype=np.uint8),
                      images=np.asarray(images[tt], dtype=np.float64))
        for a, arr in arrays.items():
            tmp = paths[tt][a] + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:               # an explicit handle: np.save would append ".npy" to the temporary name
                np.save(f, arr)
            os.replace(tmp, paths[tt][a])            # atomic: a reader sees either no file or the complete one


def _dist_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return None, 0, 1


def load_and_maybe_process_data(input_file, preprocessing_folder, force_overwrite=False, split_seed=0):
    """lidc_data_loader.py:113-136: prepare once (rank 0 only when several ranks run), then open memory-mapped."""
    paths = split_paths(preprocessing_folder)
    dist, rank, world = _dist_rank_world()
    missing = not all(os.path.exists(p) for d in paths.values() for p in d.values())
    if rank == 0 and (force_overwrite or missing):
        prepare_data(input_file, preprocessing_folder, split_seed)
    if world > 1:
        dist.barrier()                               # the other ranks open the files only after rank 0 has renamed them into place
    return {tt: {a: np.load(p, mmap_mode="r") for a, p in d.items()} for tt, d in paths.items()}
