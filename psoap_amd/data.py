"""Chunk containers, the on-disk chunk format and the Doppler-shift helpers.

Mirrors /root/reference/psoap/data.py: ``redshift`` :10-23, ``lredshift`` :25-38, ``replicate_wls``
:40-63, ``Chunk`` :120-197 (same attribute names, ``apply_mask``, ``open``/``save`` class interface)
and the file naming of /root/reference/psoap/constants.py:39.

The reference stores a chunk as ``chunk_{order}_{wl0}_{wl1}.hdf5`` with five datasets ``wl, fl, sigma,
date, mask``, each ``(n_epochs, n_pix)`` (f8 x 4 + bool).  h5py is not part of the MI355X image, so the
native container here is NumPy's ``.npz`` with exactly the same five arrays; ``Chunk.open`` prefers the
``.npz`` and reads the ``.hdf5`` only when h5py is importable (``tools/convert_chunks.py`` converts
once, on any machine that has h5py).  These are host-side helpers: the per-proposal Doppler shift of
the sampling path runs on the device (``ChunkHandle.set_grid`` + ``upload_velocities``).
"""
from __future__ import annotations

import os

import numpy as np

c_kms = 2.99792458e5                      # constants.py:12
chunk_fmt = "chunk_{:}_{:.0f}_{:.0f}"     # constants.py:39 -- order, wl0, wl1
DATASETS = ("wl", "fl", "sigma", "date", "mask")


def redshift(wl, v):
    """Relativistic Doppler shift of wavelengths; positive v lengthens (data.py:10-23)."""
    return wl * np.sqrt((c_kms + v) / (c_kms - v))


def lredshift(lwl, v):
    """Shift of ln(wavelength): ``lwl + v / c`` (data.py:25-38)."""
    return lwl + v / c_kms


def replicate_wls(lwls, velocities, mask):
    """(n_components, n_good_pix) blue-shifted copies of the masked ln-wavelength vector (data.py:40-63).

    ``lwls``: 1-D masked ln(wl) of length ``mask.sum()``; ``velocities``: (n_components, n_epochs);
    ``mask``: (n_epochs, n_pix) bool -- it routes each epoch's velocity to that epoch's good pixels.
    """
    velocities = np.asarray(velocities, dtype=np.float64)
    mask = np.asarray(mask, dtype=bool)
    n_components, n_epochs = velocities.shape
    out = np.empty((n_components, int(mask.sum())), dtype=np.float64)
    for i in range(n_components):
        out[i] = lredshift(lwls, np.broadcast_to(-velocities[i][:, None], mask.shape)[mask])
    return out


def epoch_index_of(mask):
    """Epoch of every good pixel in masked (row-major) order: what ``replicate_wls``'s broadcast encodes."""
    mask = np.asarray(mask, dtype=bool)
    return np.broadcast_to(np.arange(mask.shape[0], dtype=np.int32)[:, None], mask.shape)[mask].copy()


class Chunk:
    """One spectral chunk: ``wl, fl, sigma, date, mask``, each (n_epochs, n_pix) (data.py:120-147)."""

    def __init__(self, wl, fl, sigma, date, mask=None):
        self.wl = wl
        self.lwl = np.log(wl)
        self.fl = fl
        self.sigma = sigma
        self.date = date
        self.date1D = date[:, 0]
        self.mask = np.ones_like(self.wl, dtype=bool) if mask is None else mask
        self.n_epochs, self.n_pix = self.wl.shape

    def apply_mask(self):
        """Flatten every attribute to the good pixels (data.py:139-147); ``mask`` keeps its 2-D shape."""
        self.epoch_index = epoch_index_of(self.mask)
        self.wl = self.wl[self.mask]
        self.lwl = self.lwl[self.mask]
        self.fl = self.fl[self.mask]
        self.sigma = self.sigma[self.mask]
        self.date = self.date[self.mask]
        self.N = len(self.wl)

    @staticmethod
    def filename(order, wl0, wl1, prefix=""):
        return prefix + chunk_fmt.format(order, wl0, wl1)

    @classmethod
    def open(cls, order, wl0, wl1, limit=100, prefix=""):
        """Load ``prefix + chunk_{order}_{wl0}_{wl1}`` (.npz, else .hdf5), first ``limit`` epochs (data.py:149-172)."""
        base = cls.filename(order, wl0, wl1, prefix)
        if os.path.exists(base + ".npz"):
            with np.load(base + ".npz") as z:
                missing = [k for k in DATASETS if k not in z.files]
                if missing:
                    raise KeyError(f"{base}.npz lacks datasets {missing}")
                arrays = {k: z[k] for k in DATASETS}
        elif os.path.exists(base + ".hdf5"):
            try:
                import h5py
            except ImportError as e:
                raise ImportError(f"{base}.hdf5 needs h5py; convert it once with tools/convert_chunks.py "
                                  "on a machine that has h5py and ship the .npz") from e
            with h5py.File(base + ".hdf5", "r") as f:
                arrays = {k: f[k][:] for k in DATASETS}
        else:
            raise FileNotFoundError(f"no chunk file {base}.npz or {base}.hdf5")
        n_epochs = len(arrays["wl"])
        limit = min(int(limit), n_epochs)
        shape = arrays["wl"].shape
        for k in DATASETS:
            if arrays[k].shape != shape:
                raise ValueError(f"{base}: dataset {k} has shape {arrays[k].shape}, wl has {shape}")
        wl, fl, sigma, date = (np.asarray(arrays[k][:limit]).astype(np.float64) for k in ("wl", "fl", "sigma", "date"))
        mask = np.array(arrays["mask"][:limit], dtype=bool)
        return cls(wl, fl, sigma, date, mask)

    def save(self, order, wl0, wl1, prefix="", fmt="npz"):
        """Write the five (n_epochs, n_pix) datasets (data.py:174-197).  Only valid before ``apply_mask``."""
        if np.ndim(self.wl) != 2:
            raise ValueError("save() needs the 2-D arrays; call it before apply_mask()")
        base = self.filename(order, wl0, wl1, prefix)
        arrays = dict(wl=np.asarray(self.wl, dtype=np.float64), fl=np.asarray(self.fl, dtype=np.float64),
                      sigma=np.asarray(self.sigma, dtype=np.float64), date=np.asarray(self.date, dtype=np.float64),
                      mask=np.asarray(self.mask, dtype=bool))
        if fmt == "npz":
            np.savez(base + ".npz", **arrays)
        elif fmt == "hdf5":
            import h5py
            with h5py.File(base + ".hdf5", "w") as f:
                for k, v in arrays.items():
                    f.create_dataset(k, v.shape, dtype="bool" if k == "mask" else "f8")[:] = v
        else:
            raise ValueError("fmt must be 'npz' or 'hdf5'")
        return base + "." + fmt


def read_chunk_table(fname):
    """The ``chunks.dat`` table ``order wl0 wl1`` (/root/reference/psoap/data/chunks.dat:1, read with
    astropy's ascii reader at sample_parallel.py:58): whitespace-separated, one header line, ``#`` comments."""
    rows = []
    with open(fname) as f:
        lines = [ln.split("#", 1)[0].strip() for ln in f]
    lines = [ln for ln in lines if ln]
    if not lines:
        return rows
    header = lines[0].split()
    if header != ["order", "wl0", "wl1"]:
        raise ValueError(f"{fname}: expected header 'order wl0 wl1', got {lines[0]!r}")
    for ln in lines[1:]:
        parts = ln.split()
        if len(parts) != 3:
            raise ValueError(f"{fname}: malformed row {ln!r}")
        order = int(parts[0]) if parts[0].lstrip("+-").isdigit() else parts[0]
        rows.append((order, float(parts[1]), float(parts[2])))
    return rows


def write_chunk_table(fname, rows):
    with open(fname, "w") as f:
        f.write("order wl0 wl1\n")
        for order, wl0, wl1 in rows:
            f.write(f"{order} {wl0:.0f} {wl1:.0f}\n")


def read_mask_table(fname):
    """``masks.dat``: rows ``wl0 wl1 t0 t1`` (wavelength range x date range to reject;
    /root/reference/scripts/psoap_process_masks.py:35,63-64)."""
    rows = []
    with open(fname) as f:
        lines = [ln.split("#", 1)[0].strip() for ln in f]
    lines = [ln for ln in lines if ln]
    for ln in lines[1:]:                              # first non-comment line is the header
        parts = ln.split()
        if len(parts) != 4:
            raise ValueError(f"{fname}: malformed row {ln!r}")
        rows.append(tuple(float(x) for x in parts))
    return rows


def mask_from_regions(wl, date, regions):
    """Start from an all-good mask and reject every (wl, date) box, strict inequalities on both axes
    (psoap_process_masks.py:58-70)."""
    mask = np.ones_like(wl, dtype=bool)
    for m0, m1, t0, t1 in regions:
        mask &= ~((wl > m0) & (wl < m1) & (date > t0) & (date < t1))
    return mask


def segment_spectrum(wl, fl, sigma, date1D, order, wl0, wl1, limit=None):
    """Cut one chunk out of an (n_epochs, n_orders, n_pix) spectrum: pixels of ``order`` whose wavelength
    in the FIRST epoch lies strictly inside (wl0, wl1), first ``limit`` epochs
    (/root/reference/scripts/psoap_process_chunks.py:44-62; ``Spectrum`` date broadcast data.py:92-94)."""
    n_epochs = wl.shape[0]
    limit = n_epochs if limit is None else min(int(limit), n_epochs)
    ind = (wl[0, order, :] > wl0) & (wl[0, order, :] < wl1)
    sel = (slice(0, limit), order, ind)
    w = np.asarray(wl[sel], dtype=np.float64)
    date = np.broadcast_to(np.asarray(date1D, dtype=np.float64)[:limit, None], w.shape).copy()
    return Chunk(w, np.asarray(fl[sel], dtype=np.float64), np.asarray(sigma[sel], dtype=np.float64), date)
