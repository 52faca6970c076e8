"""LIDC-IDRI preparation - counterpart of the reference's ``data/lidc_data_loader.py``: the pickled crops
(``{key: {'image', 'masks', 'series_uid', ...}}``) are split by subject (80 / 20 test, then 80 / 20 validation, sklearn's
``train_test_split`` exactly as :72-73) and written once as arrays ``images`` (float, image - 0.5 :92), ``labels`` (uint8, H x W x 4
annotators :94-97) and ``uids`` per split.  The reference stores them as HDF5; h5py is not available in this image and the arrays
are going to live in HBM anyway, so the container is a ``.npz`` per split (memory-mapped on load).  The returned mapping has the
same ``data[tt]['images' | 'labels' | 'uids']`` surface the reference's callers index (lidc_data.py:22-53)."""
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


def prepare_data(input_file, output_folder):
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
    train_and_val_ids, split_ids["test"] = train_test_split(unique_subjects, test_size=0.2)
    split_ids["train"], split_ids["val"] = train_test_split(train_and_val_ids, test_size=0.2)
    images, labels, uids = ({tt: [] for tt in ("train", "test", "val")} for _ in range(3))
    for value in data.values():
        tt = find_subset_for_id(split_ids, value["series_uid"])
        images[tt].append(value["image"].astype(float) - 0.5)
        labels[tt].append(np.asarray(value["masks"]).transpose((1, 2, 0)))       # 4 x H x W -> H x W x 4
        uids[tt].append(hash(value["series_uid"]))
    os.makedirs(output_folder, exist_ok=True)
    for tt in ("test", "train", "val"):
        np.savez(os.path.join(output_folder, f"data_lidc_{tt}.npz"), uids=np.asarray(uids[tt], dtype=np.int64),
                 labels=np.asarray(labels[tt], dtype=np.uint8), images=np.asarray(images[tt], dtype=np.float64))


def load_and_maybe_process_data(input_file, preprocessing_folder, force_overwrite=False):
    """lidc_data_loader.py:113-136: prepare once, then open."""
    paths = {tt: os.path.join(preprocessing_folder, f"data_lidc_{tt}.npz") for tt in ("train", "test", "val")}
    if force_overwrite or not all(os.path.exists(p) for p in paths.values()):
        prepare_data(input_file, preprocessing_folder)
    return {tt: np.load(p, mmap_mode="r") for tt, p in paths.items()}
