"""psoap_amd -- MI355X (gfx950) native GP-likelihood hot path of PSOAP.

``covariance`` and ``matrix_functions`` mirror the reference modules of the same names (likelihoods,
predict family, calibration family, fills); ``chunk`` is the batched / resident API; ``lnprob``,
``orbit`` and ``utils`` form the ``lnprob(p)`` boundary with the orbit solve on the device;
``samplers``, ``priors``, ``sample_parallel`` and ``ensemble`` are the multi-chain, multi-chunk,
multi-GPU sampling layer; ``data`` holds the chunk container and file formats.  The compute path is
the HIP library behind ``include/psoap_gp.h`` (built by ``psoap_amd.build``); there is no CPU fallback.
"""
__version__ = "0.1.0"
