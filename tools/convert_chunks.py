"""One-time conversion of PSOAP chunk files between .hdf5 (reference, needs h5py) and .npz (this package).

    python tools/convert_chunks.py chunk_22_5160_5170.hdf5 [...]      # -> .npz next to each input
"""
import os
import sys

import numpy as np

DATASETS = ("wl", "fl", "sigma", "date", "mask")


def main(paths):
    try:
        import h5py
    except ImportError:
        sys.exit("h5py is required for the conversion; run this where the reference itself runs")
    for p in paths:
        base, ext = os.path.splitext(p)
        if ext == ".hdf5":
            with h5py.File(p, "r") as f:
                arrays = {k: f[k][:] for k in DATASETS}
            arrays["mask"] = np.asarray(arrays["mask"], dtype=bool)
            np.savez(base + ".npz", **arrays)
            print(p, "->", base + ".npz")
        elif ext == ".npz":
            with np.load(p) as z, h5py.File(base + ".hdf5", "w") as f:
                for k in DATASETS:
                    f.create_dataset(k, z[k].shape, dtype="bool" if k == "mask" else "f8")[:] = z[k]
            print(p, "->", base + ".hdf5")
        else:
            sys.exit(f"unknown extension: {p}")


if __name__ == "__main__":
    main(sys.argv[1:])
