"""Synthetic radial feeders and Monte-Carlo load scenarios (benchmark inputs).

The reference ships no large network.  This generator is the one specified in
SURVEY.md Appendix E (survey-authored input spec, not reference code): it only
writes bus/line CSVs in the reference's net2 dialect
(`Harmonic Power Flow/net2_buses.csv:1`, `net2_lines.csv:1`), so the files can be
fed both to the reference (to capture goldens) and to this package.

NumPy `default_rng(seed)` (PCG64) call order matters and must not change:
one `permutation`, then n-1 `integers` (parents), then one `integers` per bus
(load), then one `integers` per line (impedance).
"""
import os

import numpy as np

_RX = [(0.5, 0.5), (1, 4), (0.5, 1), (0.5, 1), (0.5, 1)]          # ohm, palette of net1_lines.csv
_PQ = [(100, 100), (100, 100), (150, 100), (250, 100), (0, 0)]    # W/var, palette of net1_buses.csv


def gen(n, seed=0, frac_nl=0.35, zscale=None, prefix="syn", outdir="."):
    """Write `<prefix><n>_buses.csv` / `_lines.csv` into `outdir`; return the two paths."""
    rng = np.random.default_rng(seed)
    n_nl = int(round(frac_nl * n))
    n_lin = n - n_nl                      # IDs 1..n_lin: slack+PQ ; n_lin+1..n: nonlinear
    zscale = zscale if zscale is not None else 20.0 / n
    order = np.concatenate([[1], 1 + rng.permutation(np.arange(1, n))])   # placement order, slack first
    parent = {}
    for pos in range(1, n):
        parent[order[pos]] = order[rng.integers(0, pos)]                  # random recursive tree
    fb = os.path.join(outdir, f"{prefix}{n}_buses.csv")
    fl = os.path.join(outdir, f"{prefix}{n}_lines.csv")
    with open(fb, "w") as f:
        f.write("ID;type;component;S;P;Q;X_sh\n1;slack;generator;0;0;0;0.005\n")
        for i in range(2, n + 1):
            p, q = _PQ[rng.integers(0, len(_PQ))]
            if i <= n_lin:
                f.write(f"{i};PQ;lin_load_{i};0;{p};{q};0\n")
            else:
                f.write(f"{i};nonlinear;smps;0;{p};{q};0\n")
    with open(fl, "w") as f:
        f.write("ID;fromID;toID;R;X;G;B\n")
        for k, (child, par) in enumerate(sorted(parent.items()), start=1):
            r, x = _RX[rng.integers(0, len(_RX))]
            f.write(f"{k};{par};{child};{r * zscale:.10g};{x * zscale:.10g};0;0\n")
    return fb, fl


def scenario_scale(n, s):
    """Per-bus load multiplier of Monte-Carlo scenario `s`:
    u ~ U[0.5, 1.5] i.i.d. per bus from `default_rng(1000 + s)`; P_i,Q_i <- P_i,Q_i * u_i."""
    return np.random.default_rng(1000 + s).uniform(0.5, 1.5, size=n)


def add_ties(fl, n, k, seed=42):
    """Append k loop-closing lines to a lines CSV written by `gen` (random bus pairs that are not yet connected, impedances from the feeder's palette):
    the meshed variants of the synthetic feeders (the networks of the meshed oracle fixtures: same generator, same seed) -> list of (from, to)."""
    rows = open(fl).read().splitlines()
    have = set()
    for r in rows[1:]:
        c = r.split(";")
        have.add((min(int(c[1]), int(c[2])), max(int(c[1]), int(c[2]))))
    rng = np.random.default_rng(seed)
    pal = [(0.5, 0.5), (1, 4), (0.5, 1)]
    out = []
    lid = len(rows)
    while len(out) < k:
        a, b = int(rng.integers(2, n + 1)), int(rng.integers(2, n + 1))
        if a == b or (min(a, b), max(a, b)) in have:
            continue
        have.add((min(a, b), max(a, b)))
        r, x = pal[int(rng.integers(0, len(pal)))]
        rows.append("%d;%d;%d;%.10g;%.10g;0;0" % (lid, a, b, r * 20.0 / n, x * 20.0 / n))
        lid += 1
        out.append((a, b))
    open(fl, "w").write("\n".join(rows) + "\n")
    return out
