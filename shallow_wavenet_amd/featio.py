"""On-disk feature and statistics formats (row f3 of SURVEY.md section 8).

The reference keeps per-utterance features and the corpus statistics in HDF5 files of named datasets
(`src/utils/utils.py:38-126` read_hdf5 / write_hdf5 / shape_hdf5 / check_hdf5):
  * `<utt>.h5`  dataset `/feat_org_lf0`: (Tf, n_aux) float64 rows `[uv, log f0, codeap.., mcep..]`
    (`feature_extract.py:309-312`), read by the training generator and the decode drivers;
  * `stats.h5`  datasets `/mean_org_lf0`, `/scale_org_lf0` (`calc_stats.py:204-209`; for another `--string_path`
    the names are `/mean_<string_path>` and `/scale_<string_path>`), read into `scale_in` by the training scripts
    (`train_cswnv_laplace-stftcmplx_shift1.py:316-350`).
h5py is not importable in the build image, so every function here speaks two container formats behind one interface:
  * `.h5`  - through h5py when it is importable (RuntimeError otherwise);
  * `.npz` - the same named datasets as arrays of an uncompressed numpy archive (key = dataset name without the
    leading "/"); `.npy` - one anonymous dataset (per-utterance feature files only).
Nothing is ever unpickled (`allow_pickle=False`).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np


def _h5py():
    try:
        import h5py  # noqa: WPS433  (optional dependency, absent in the build image)
        return h5py
    except ImportError as e:
        raise RuntimeError("HDF5 files need h5py; use the .npz / .npy side format or install h5py") from e


def _key(dataset: str) -> str:
    return dataset.lstrip("/")


def check_dataset(path: str, dataset: str) -> bool:
    """utils.py:16-35 check_hdf5: does `dataset` exist in the file?"""
    if not os.path.exists(path):
        return False
    if path.endswith(".h5"):
        with _h5py().File(path, "r") as f:
            return dataset in f
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return _key(dataset) in z.files
    return path.endswith(".npy")


def read_dataset(path: str, dataset: str = "/feat_org_lf0") -> np.ndarray:
    """utils.py:38-64 read_hdf5: the dataset's values."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"There is no such a feature / stats file. ({path})")
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=False)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            if _key(dataset) not in z.files:
                raise KeyError(f"There is no such a data in the file. ({dataset})")
            return z[_key(dataset)]
    if path.endswith(".h5"):
        with _h5py().File(path, "r") as f:
            if dataset not in f:
                raise KeyError(f"There is no such a data in hdf5 file. ({dataset})")
            return f[dataset][()]
    raise RuntimeError(f"unsupported container {path}")


def dataset_shape(path: str, dataset: str = "/feat_org_lf0") -> Tuple[int, ...]:
    """utils.py:67-83 shape_hdf5, without reading the data where the container allows it."""
    if path.endswith(".npy"):
        return tuple(np.load(path, mmap_mode="r", allow_pickle=False).shape)
    if path.endswith(".h5"):
        with _h5py().File(path, "r") as f:
            return tuple(f[dataset].shape)
    return tuple(read_dataset(path, dataset).shape)


def write_dataset(path: str, dataset: str, data, is_overwrite: bool = True) -> int:
    """utils.py:86-126 write_hdf5: create the folder, add (or replace) one named dataset, keep the others."""
    data = np.array(data)
    folder = os.path.dirname(path)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    if path.endswith(".npy"):
        np.save(path, data, allow_pickle=False)
        return 1
    if path.endswith(".npz"):
        have: Dict[str, np.ndarray] = {}
        if os.path.exists(path):
            with np.load(path, allow_pickle=False) as z:
                have = {k: z[k] for k in z.files}
        if _key(dataset) in have and not is_overwrite:
            raise RuntimeError("there is already dataset. if you want to overwrite, please set is_overwrite = True.")
        have[_key(dataset)] = data
        tmp = path + ".tmp.npz"
        np.savez(tmp, **have)
        os.replace(tmp, path)
        return 1
    if path.endswith(".h5"):
        h5py = _h5py()
        with h5py.File(path, "r+" if os.path.exists(path) else "w") as f:
            if dataset in f:
                if not is_overwrite:
                    raise RuntimeError("there is already dataset. if you want to overwrite, please set is_overwrite = True.")
                del f[dataset]
            f.create_dataset(dataset, data=data)
        return 1
    raise RuntimeError(f"unsupported container {path}")


def side_path(path: str, ext: str = ".npz") -> str:
    """`x.h5` -> `x.npz`: where a list written for the reference (HDF5 names) finds the side-format file."""
    return os.path.splitext(path)[0] + ext


def resolve(path: str) -> str:
    """the file to open for a list entry: itself if it exists, else its .npz / .npy sibling."""
    if os.path.exists(path):
        return path
    for ext in (".npz", ".npy", ".h5"):
        if os.path.exists(side_path(path, ext)):
            return side_path(path, ext)
    return path


# ----------------------------------------------------------------------------------------- statistics
class RunningStats:
    """per-dimension mean / population standard deviation accumulated utterance by utterance - the arithmetic of
    sklearn's `StandardScaler.partial_fit` that calc_stats.py:150-198 runs over every feature file (Chan et al.
    pairwise update of the sum of squared deviations; a zero-variance dimension gets scale 1.0)."""

    def __init__(self):
        self.n = 0
        self.mean_: Optional[np.ndarray] = None
        self._m2: Optional[np.ndarray] = None

    def partial_fit(self, x) -> "RunningStats":
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 2:
            raise ValueError("features must be (frames, dims)")
        nb = x.shape[0]
        if nb == 0:
            return self
        mb = x.mean(0)
        m2b = ((x - mb) ** 2).sum(0)
        if self.n == 0:
            self.n, self.mean_, self._m2 = nb, mb, m2b
            return self
        tot = self.n + nb
        delta = mb - self.mean_
        self._m2 = self._m2 + m2b + delta ** 2 * (self.n * nb / tot)
        self.mean_ = self.mean_ + delta * (nb / tot)
        self.n = tot
        return self

    @property
    def var_(self) -> np.ndarray:
        return self._m2 / self.n

    @property
    def scale_(self) -> np.ndarray:
        s = np.sqrt(self.var_)
        # sklearn _handle_zeros_in_scale: constant features are not scaled
        s[s < 10 * np.finfo(s.dtype).eps] = 1.0
        return s


def stats_names(string_path: str) -> Tuple[str, str]:
    """dataset names calc_stats.py:204-209 writes."""
    if string_path == "/feat_org_lf0":
        return "/mean_org_lf0", "/scale_org_lf0"
    return "/mean_" + string_path, "/scale_" + string_path


def calc_stats(feat_files: Iterable[str], string_path: str = "/feat_org_lf0") -> Tuple[np.ndarray, np.ndarray]:
    """mean / scale of `string_path` over a list of feature files (calc_stats.py:150-200)."""
    st = RunningStats()
    for f in feat_files:
        st.partial_fit(read_dataset(resolve(f), string_path))
    if st.n == 0:
        raise RuntimeError("no feature frames found")
    return st.mean_, st.scale_


def write_stats(path: str, string_path: str, mean, scale) -> None:
    mname, sname = stats_names(string_path)
    write_dataset(path, mname, np.asarray(mean, dtype=np.float64))
    write_dataset(path, sname, np.asarray(scale, dtype=np.float64))


def read_stats(path: str, string_path: str = "/feat_org_lf0") -> Tuple[np.ndarray, np.ndarray]:
    """the three-way name lookup of the training scripts (train_cswnv...py:316-329):
    /mean_<name> | /mean_<string_path> | /mean_feat_<name>, with <name> = string_path after "feat_"."""
    path = resolve(path)
    name = string_path.split("feat_")[1]
    for m in ("/mean_" + name, "/mean_" + string_path, "/mean_feat_" + name):
        if check_dataset(path, m):
            return np.asarray(read_dataset(path, m)), np.asarray(read_dataset(path, m.replace("mean", "scale", 1)))
    # round-1 side files: plain "mean" / "scale" arrays
    if path.endswith(".npz") and check_dataset(path, "mean"):
        return np.asarray(read_dataset(path, "mean")), np.asarray(read_dataset(path, "scale"))
    raise KeyError(f"{path}: no statistics for {string_path}")


def main(argv: Optional[Sequence[str]] = None) -> int:
    """counterpart of src/bin/calc_stats.py (run.sh stage 2): --feats list, --stats output, --string_path."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--feats", required=True, help="name of the list of feature files")
    p.add_argument("--wavs", default=None, help="accepted for compatibility (unused by the reference too)")
    p.add_argument("--n_quantize", default=256, type=int)
    p.add_argument("--string_path", default="/feat_org_lf0", type=str, help="dataset name")
    p.add_argument("--stats", required=True, help="output file (.h5 with h5py, else .npz)")
    args = p.parse_args(argv)
    with open(args.feats) as f:
        files = [ln.strip() for ln in f if ln.strip()]
    print("number of training utterances =", len(files))
    print(args.string_path)
    mean, scale = calc_stats(files, args.string_path)
    print(mean)
    print(scale)
    out = args.stats
    if out.endswith(".h5"):
        try:
            _h5py()
        except RuntimeError:
            out = side_path(out)
            print(f"h5py is not importable: writing {out}")
    write_stats(out, args.string_path, mean, scale)
    return 0
