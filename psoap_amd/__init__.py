"""psoap_amd -- MI355X (gfx950) native GP-likelihood hot path of PSOAP.

``psoap_amd.covariance`` and ``psoap_amd.matrix_functions`` mirror the reference
modules of the same names for the dense likelihood / prediction path;
``psoap_amd.chunk`` and ``psoap_amd.ensemble`` add the batched and multi-GPU
entry points.  The compute path is the HIP library behind ``include/psoap_gp.h``
(built by ``psoap_amd.build``); there is no CPU fallback.
"""
__version__ = "0.1.0"
