"""End to end on synthetic data: write chunk files and a config the way a PSOAP working directory holds
them, sample with B Metropolis-Hastings chains on the GPU, check convergence, reconstruct the component
spectra.  (Needs an MI355X and the built library: python -m psoap_amd.build.)

    python examples/quickstart.py [workdir] [--chunks 3] [--chains 8] [--samples 60]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from psoap_amd import data as pdata                     # noqa: E402
from psoap_amd import retrieve, samplers, utils         # noqa: E402
from psoap_amd import sample_parallel as sp             # noqa: E402
from psoap_amd import synthetic as syn                  # noqa: E402

PARAMETERS = dict(q=0.6, K=25.0, e=0.1, omega=30.0, P=12.0, T0=2455010.0, gamma=3.0,
                  amp_f=0.2, l_f=6.0, amp_g=0.1, l_g=8.0)
JUMPS = dict(q=0.004, K=0.1, e=0.004, omega=0.3, P=0.004, T0=0.02, gamma=0.01,
             amp_f=0.004, l_f=0.1, amp_g=0.004, l_g=0.1)


def write_dataset(workdir, n_chunks, n_epochs=10, n_pix=80):
    """chunk_*.npz + chunks.dat + config.yaml: what the reference's drivers read from their CWD."""
    import yaml
    rows = []
    for k in range(n_chunks):
        s = syn.make_chunk(2, n_epochs, n_pix, seed=100 + k, masked_fraction=0.05)

        def full(v, fill):
            out = np.full(s.mask.shape, fill)
            out[s.mask] = v
            return out
        date = np.broadcast_to(s.dates[:, None], s.mask.shape).copy()
        pdata.Chunk(np.exp(full(s.lwl, 8.5)), full(s.fl, 1.0), full(s.sigma, 1.0), date, s.mask).save(
            20 + k, 5100.0 + 10 * k, 5110.0 + 10 * k, prefix=workdir + "/")
        rows.append((20 + k, 5100.0 + 10 * k, 5110.0 + 10 * k))
    pdata.write_chunk_table(os.path.join(workdir, "chunks.dat"), rows)
    config = dict(model="SB2", chunk_file=os.path.join(workdir, "chunks.dat"), epoch_limit=n_epochs, soften=1.0,
                  parameters=PARAMETERS, jumps=JUMPS, fix_params=["gamma"], samples=60,
                  opt_jump=os.path.join(workdir, "opt_jump.npy"), outdir=os.path.join(workdir, "output"))
    with open(os.path.join(workdir, "config.yaml"), "w") as f:
        yaml.safe_dump(config, f)
    return config


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("workdir", nargs="?", default="psoap_quickstart")
    ap.add_argument("--chunks", type=int, default=3)
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--samples", type=int, default=60)
    args = ap.parse_args(argv)
    os.makedirs(args.workdir, exist_ok=True)
    config = write_dataset(args.workdir, args.chunks)
    config["samples"] = args.samples + args.samples % 2
    chunks = sp.load_chunks(config, prefix=args.workdir + "/")
    print("chunks:", [c.N for c in chunks], "pixels;", args.chains, "chains x", config["samples"], "samples")
    sampler = sp.run(config, chunks, run_index=0, n_chains=args.chains, seed=1)
    chains = [np.load(os.path.join(config["outdir"], "run{:02d}".format(b), "flatchain.npy")) for b in range(args.chains)]
    mean, std, rhat = samplers.gelman_rubin(chains)
    print("R_hat:", np.round(rhat, 2))
    np.save(config["opt_jump"], utils.estimate_covariance(np.concatenate(chains)))       # proposal of the next run
    pars = dict(PARAMETERS)
    pars.update(dict(zip([n for n in utils.registered_params["SB2"] if n != "gamma"], mean)))
    order, wl0, wl1 = pdata.read_chunk_table(config["chunk_file"])[0]
    res = retrieve.retrieve_components("SB2", pdata.Chunk.open(order, wl0, wl1, prefix=args.workdir + "/"), pars)
    retrieve.save_components(res, os.path.join(args.workdir, "plots_" + pdata.chunk_fmt.format(order, wl0, wl1)))
    print("reconstructed f, g on", len(res["wl_predict"]), "pixels; acceptance", np.round(sampler.acceptance_fraction, 2))
    return sampler, res


if __name__ == "__main__":
    main()
