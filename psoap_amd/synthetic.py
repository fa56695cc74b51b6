"""Seeded synthetic spectra for the GP-likelihood hot path.

The reference ships no spectra, so every test, golden fixture and bench line is
driven by this generator.  It restates the *data conventions* of the reference,
nothing more:

* a chunk is ``n_epochs x n_pix`` ln-wavelength / flux / sigma arrays flattened
  epoch-major, exactly as ``Chunk.apply_mask`` produces them
  (/root/reference/psoap/data.py:138-147);
* per-component rest-frame grids are ``lwl - v[c, epoch] / c_kms``
  (``replicate_wls`` + ``lredshift``, /root/reference/psoap/data.py:25-63).

Everything is pure NumPy with ``numpy.random.default_rng(seed)`` so the very
same arrays are regenerated on the GPU box from the seed alone.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

C_KMS = 2.99792458e5  # km/s; /root/reference/psoap/constants.py:13

# GP hyper-parameters of the benchmark parameter set (amp in flux, l in km/s)
GP_BASE = {
    1: (0.2, 5.0),
    2: (0.2, 5.0, 0.1, 7.0),
    3: (0.2, 5.0, 0.1, 7.0, 0.05, 6.0),
}
# Metropolis step sizes of the packaged config (amp, l) --
# /root/reference/psoap/data/config.SB2.yaml:42-45
GP_JUMP = (0.05, 0.5)

# BASELINE.json configs -> (n_components, n_epochs, n_pix)
CONFIG_SHAPES = {
    1: (1, 10, 200),   # SB1 N=2000
    2: (1, 16, 256),   # SB1 N=4096
    3: (2, 20, 300),   # SB2 N=6000  (the metric's unit of work)
    4: (2, 20, 300),   # SB2 ensemble: 32 walkers x 8 chunks of cfg3 shape
    5: (3, 16, 512),   # ST3 N=8192
}


@dataclass
class SyntheticChunk:
    n_components: int
    n_epochs: int
    n_pix: int
    lwl: np.ndarray         # (N,) observed-frame ln-wavelengths, epoch-major
    velocities: np.ndarray  # (c, n_epochs) km/s
    lwls: np.ndarray        # (c, N) rest-frame ln-wavelengths per component
    fl: np.ndarray          # (N,)
    sigma: np.ndarray       # (N,)
    seed: int
    mask: np.ndarray        # (n_epochs, n_pix) bool; N = mask.sum()
    dates: np.ndarray = None  # (n_epochs,) observation dates [JD], for the orbit / lnprob(p) boundary

    @property
    def N(self) -> int:
        return self.lwl.shape[0]

    @property
    def epoch_index(self) -> np.ndarray:
        """(N,) epoch of every unmasked pixel (what the mask broadcast in
        ``replicate_wls`` encodes)."""
        return epoch_index_from_mask(self.mask)


def epoch_index_from_mask(mask: np.ndarray) -> np.ndarray:
    n_epochs, n_pix = mask.shape
    return np.repeat(np.arange(n_epochs), n_pix).reshape(mask.shape)[mask]


def replicate_wls(lwl: np.ndarray, velocities: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """Rest-frame grids ``lwl - v[c, epoch]/c_kms`` for the unmasked pixels.

    Same arithmetic as ``replicate_wls`` -> ``lredshift(lwls, -v)``
    (/root/reference/psoap/data.py:37,61): ``lwl + (-v)/c_kms``.
    """
    c, n_epochs = velocities.shape
    ep = epoch_index_from_mask(mask)
    out = np.empty((c, lwl.shape[0]), dtype=np.float64)
    for i in range(c):
        out[i] = lwl + (-velocities[i][ep]) / C_KMS
    return out


def make_dates(n_epochs: int, seed: int) -> np.ndarray:
    """Sorted observation dates [JD] from an independent stream (does not disturb the chunk's draws)."""
    rng = np.random.default_rng([seed, 0x0DA7E5])
    return np.sort(rng.uniform(2455000.0, 2455400.0, size=n_epochs))


# orbital parameters (registered_params order up to gamma) used by fixtures and examples
ORBIT_BASE = {
    "SB1": (12.0, 0.25, 40.0, 23.0, 2455010.0, 3.0),
    "SB2": (0.6, 12.0, 0.25, 40.0, 23.0, 2455010.0, 3.0),
    "ST1": (12.0, 0.25, 40.0, 23.0, 2455010.0, 4.0, 0.1, 200.0, 310.0, 2455100.0, 3.0),
    "ST2": (0.6, 12.0, 0.25, 40.0, 23.0, 2455010.0, 4.0, 0.1, 200.0, 310.0, 2455100.0, 3.0),
    "ST3": (0.6, 12.0, 0.25, 40.0, 23.0, 2455010.0, 0.3, 4.0, 0.1, 200.0, 310.0, 2455100.0, 3.0),
}


def make_orbit_proposals(model: str, n: int, seed: int) -> np.ndarray:
    """(n, n_orb) orbital parameter vectors: the base vector (row 0) plus seeded perturbations with
    eccentricities kept in [0, 0.9)."""
    rng = np.random.default_rng([seed, 0x0B17])
    base = np.array(ORBIT_BASE[model], dtype=np.float64)
    out = np.repeat(base[None], n, axis=0)
    scale = np.maximum(np.abs(base) * 0.1, 0.05)
    out[1:] += scale * rng.standard_normal(out[1:].shape)
    names_e = {"SB1": [1], "SB2": [2], "ST1": [1, 6], "ST2": [2, 7], "ST3": [2, 8]}[model]
    for j in names_e:
        out[:, j] = np.clip(np.abs(out[:, j]), 0.0, 0.89)
    names_pos = {"SB1": [0, 3], "SB2": [0, 1, 4], "ST1": [0, 3, 5, 8], "ST2": [0, 1, 4, 6, 9],
                 "ST3": [0, 1, 4, 6, 7, 10]}[model]
    for j in names_pos:
        out[:, j] = np.abs(out[:, j]) + 1e-3
    return out


def _line_list(rng, lo, hi, n_pix, ratio):
    n_lines = max(1, n_pix // 40)
    centres = rng.uniform(lo, hi, size=n_lines)
    depths = rng.uniform(0.05, 0.5, size=n_lines) * ratio
    return centres, depths


def _template(x, centres, depths, width_kms=6.0):
    w = width_kms / C_KMS
    out = np.zeros_like(x)
    for c0, d in zip(centres, depths):
        out -= d * np.exp(-0.5 * ((x - c0) / w) ** 2)
    return out


def make_chunk(n_components: int, n_epochs: int, n_pix: int, seed: int,
               realistic_flux: bool = True, sigma0: float = 0.02,
               masked_fraction: float = 0.0) -> SyntheticChunk:
    """One synthetic chunk (SURVEY.md section 8(d) recipe).

    ``masked_fraction`` > 0 drops that fraction of pixels at random (ragged
    epochs), the way ``Chunk.apply_mask`` does.
    """
    rng = np.random.default_rng(seed)
    delta = 2.7 / C_KMS
    lwl0 = np.log(5200.0) + np.arange(n_pix) * delta
    jitter = rng.uniform(-0.5, 0.5, size=n_epochs) * delta
    velocities = rng.uniform(-60.0, 60.0, size=(n_components, n_epochs))
    if masked_fraction > 0.0:
        mask = rng.uniform(size=(n_epochs, n_pix)) >= masked_fraction
    else:
        mask = np.ones((n_epochs, n_pix), dtype=bool)
    lwl = (lwl0[None, :] + jitter[:, None])[mask]
    lwls = replicate_wls(lwl, velocities, mask)
    if realistic_flux:
        ratios = (1.0, 0.4, 0.2)
        fl = np.ones_like(lwl)
        lo, hi = lwls.min(), lwls.max()
        for c in range(n_components):
            centres, depths = _line_list(rng, lo, hi, n_pix, ratios[c])
            fl = fl + _template(lwls[c], centres, depths)
        fl = fl + sigma0 * rng.standard_normal(lwl.shape[0])
    else:
        fl = 1.0 + 0.05 * rng.standard_normal(lwl.shape[0])
    sigma = np.full(lwl.shape[0], sigma0, dtype=np.float64)
    return SyntheticChunk(n_components, n_epochs, n_pix,
                          np.ascontiguousarray(lwl), velocities,
                          np.ascontiguousarray(lwls), np.ascontiguousarray(fl),
                          sigma, seed, mask, make_dates(n_epochs, seed))


def make_config_chunk(cfg: int, chunk_index: int = 0, **kw) -> SyntheticChunk:
    """Chunk for BASELINE.json config ``cfg`` with ``seed = 1000*cfg + chunk_index``."""
    c, n_epochs, n_pix = CONFIG_SHAPES[cfg]
    return make_chunk(c, n_epochs, n_pix, seed=1000 * cfg + chunk_index, **kw)


def make_walkers(n_components: int, n_walkers: int, seed: int) -> np.ndarray:
    """(n_walkers, 2c) GP parameter vectors: base + Gaussian jumps, all positive.

    Walker 0 is the unperturbed base vector.
    """
    rng = np.random.default_rng(seed)
    base = np.array(GP_BASE[n_components], dtype=np.float64)
    step = np.tile(np.array(GP_JUMP), n_components)
    out = np.empty((n_walkers, base.size))
    out[0] = base
    i = 1
    while i < n_walkers:
        p = base + step * rng.standard_normal(base.size)
        if np.all(p > 0.0):
            out[i] = p
            i += 1
    return out


def make_walker_velocities(chunk: SyntheticChunk, n_walkers: int, seed: int,
                           scale_kms: float = 0.2) -> np.ndarray:
    """(n_walkers, c, n_epochs) velocity tables: the chunk's velocities plus small
    seeded perturbations (what an orbit proposal does to the Doppler shifts)."""
    rng = np.random.default_rng(seed)
    v = np.repeat(chunk.velocities[None], n_walkers, axis=0).copy()
    v[1:] += scale_kms * rng.standard_normal(v[1:].shape)
    return v


def walker_lwls(chunk: SyntheticChunk, walker_velocities: np.ndarray) -> np.ndarray:
    """(n_walkers, c, N) rest-frame grids for each walker's velocity table."""
    return np.stack([replicate_wls(chunk.lwl, v, chunk.mask) for v in walker_velocities])
