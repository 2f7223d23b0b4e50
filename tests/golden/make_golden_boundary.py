#!/usr/bin/env python3
"""
Golden fixture for the reference's threshold comparisons AT their boundaries (boundary_diag.npz): states placed so that a norm
equals a limit, or sits one or two units in the last place either side of it, are written into the UNMODIFIED reference env (same
inert import stubs as make_golden.py) the way monte_carlo.py:107-112 does, and its own helper methods are asked:
get_errors() (:451), check_collision() (:388), check_success() (:406), dist_from_koz() (:510).

Limits covered: max_rd_error (position, `<=` :416), max_vd_error (velocity), max_wd_error (rotation rate), koz_radius (`<` :397).
With both attitudes at identity the goal position is exactly [0, -2, 0] and the error vectors have one non-zero component, so the
norm the reference takes is exactly that component: the fixture pins which side of each limit `<=` / `<` puts the boundary value.

    python tests/golden/make_golden_boundary.py      # seconds; needs /root/reference (build container only)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, diag8, install_stubs   # noqa: E402


def around(x, k=2):
    out = [x]
    lo = hi = x
    for _ in range(k):
        lo, hi = np.nextafter(lo, -np.inf), np.nextafter(hi, np.inf)
        out += [lo, hi]
    return sorted(out)


def cases(env):
    base = np.zeros(20)
    base[6] = 1.0; base[13] = 1.0
    out = []
    for x in around(env.max_rd_error):                       # position error = x exactly: rc = goal + [0, -x, 0]
        s = base.copy(); s[1] = -2.0 - x
        out.append(("pos", x, s))
    for x in around(env.max_vd_error):                       # velocity error = |vc| (target at rest)
        s = base.copy(); s[1] = -2.25; s[3] = x
        out.append(("vel", x, s))
    for x in around(env.max_wd_error):                       # rotation-rate error = |wc| (target at rest)
        s = base.copy(); s[1] = -2.25; s[10] = x
        out.append(("rot", x, s))
    for x in around(env.koz_radius):                         # |rc| = x, 90 deg off the corridor axis: collision iff inside the sphere
        s = base.copy(); s[0] = x
        out.append(("koz", x, s))
    # generic directions: the norm is a rounded sum of three squares; scale a fixed direction so that it lands on / next to the limit
    rng = np.random.default_rng(7)
    for name, limit, lo in (("pos3", env.max_rd_error, 0), ("vel3", env.max_vd_error, 3), ("rot3", env.max_wd_error, 10)):
        for _ in range(6):
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            for x in around(limit, 1):
                v = u * x
                s = base.copy(); s[1] = -2.0 if name == "pos3" else -2.25
                if name == "pos3":
                    s[0:3] = np.array([0.0, -2.0, 0.0]) + v
                else:
                    s[lo:lo + 3] = v
                out.append((name, x, s))
    return out


def main():
    install_stubs()
    from rendezvous_env import RendezvousEnv
    env = RendezvousEnv(quiet=True)
    np.random.seed(0)
    env.reset()
    cs = cases(env)
    n = len(cs)
    rec = dict(kind=np.array([c[0] for c in cs]), value=np.array([c[1] for c in cs]), state=np.stack([c[2] for c in cs]),
               diag=np.zeros((n, 8)))
    for i, (_, _, s) in enumerate(cs):
        env.rc, env.vc, env.qc = s[0:3].copy(), s[3:6].copy(), s[6:10].copy()
        env.wc, env.qt, env.wt = s[10:13].copy(), s[13:17].copy(), s[17:20].copy()
        env.collided = False
        rec["diag"][i] = diag8(env)
    np.savez_compressed(os.path.join(OUT, "boundary_diag.npz"), **rec)
    for k in ("pos", "vel", "rot", "koz"):
        sel = rec["kind"] == k
        print(k, [(float(v), int(d[5]), int(d[4])) for v, d in zip(rec["value"][sel], rec["diag"][sel])])
    print(n, "cases; success flags set:", int(rec["diag"][:, 5].sum()), "collision flags set:", int(rec["diag"][:, 4].sum()))


if __name__ == "__main__":
    main()
