"""CPU oracle for the orbit models -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py for the rules).

NumPy restatement of /root/reference/psoap/orbit.py: true anomaly :47-72 / :213-254, velocity
terms :74-82, :135-139, :256-266, :349-353, :443-447 and the ``get_velocities`` assemblies.  The
reference solves Kepler's equation with ``fsolve`` per date; this restatement iterates Newton's
method to machine precision.  Pinned against velocities produced by the reference's own classes
(tests/golden/make_golden.py -> golden_orbit_v1.npz) to 1e-8 km/s.
"""
import numpy as np

C_KMS = 2.99792458e5


def true_anomaly(t, T0, P, e):
    tt = np.mod(np.asarray(t, dtype=np.float64) - T0, P)
    M = 2 * np.pi * tt / P
    E = M.copy() if e < 0.8 else np.full_like(M, np.pi)
    for _ in range(100):
        dE = (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
        E = E - dE
        if np.max(np.abs(dE)) <= 1e-16:
            break
    th = 2 * np.arctan(np.sqrt((1 + e) / (1 - e)) * np.tan(E / 2.0))
    return np.where(E < np.pi, th, th + 2 * np.pi)


def _term(K, e, omega_deg, f):
    w = omega_deg * np.pi / 180
    return K * (np.cos(w + f) + e * np.cos(w))


def velocities(model, p, dates):
    """(c, n_dates) km/s for one orbital parameter vector in registered_params order up to gamma."""
    p = [float(x) for x in p]
    if model == "SB1":
        K, e, om, P, T0, g = p
        f = true_anomaly(dates, T0, P, e)
        return np.atleast_2d(_term(K, e, om, f) + g)
    if model == "SB2":
        q, K, e, om, P, T0, g = p
        f = true_anomaly(dates, T0, P, e)
        return np.vstack((_term(K, e, om, f) + g, _term(K / q, e, om + 180, f) + g))
    if model == "ST1":
        K_in, e_in, om_in, P_in, T0_in, K_out, e_out, om_out, P_out, T0_out, g = p
        q_in = q_out = None
    elif model == "ST2":
        q_in, K_in, e_in, om_in, P_in, T0_in, K_out, e_out, om_out, P_out, T0_out, g = p
        q_out = None
    else:
        q_in, K_in, e_in, om_in, P_in, T0_in, q_out, K_out, e_out, om_out, P_out, T0_out, g = p
    f_in = true_anomaly(dates, T0_in, P_in, e_in)
    f_out = true_anomaly(dates, T0_out, P_out, e_out)
    v3 = _term(K_out, e_out, om_out, f_out)
    rows = [_term(K_in, e_in, om_in, f_in) + v3 + g]
    if q_in is not None:
        rows.append(_term(K_in / q_in, e_in, om_in + 180, f_in) + v3 + g)
    if q_out is not None:
        rows.append(_term(K_out / q_out, e_out, om_out + 180, f_out) + g)
    return np.vstack(rows)
