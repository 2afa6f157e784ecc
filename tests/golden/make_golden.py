#!/usr/bin/env python3
"""Generate the committed golden vectors from the ORACLE (the CPU restatement of the reference algorithm).

The reference itself is pure Julia and cannot run in this pipeline (no Julia runtime here or on the GPU box), and
its repository holds no binary fixtures for this path — so these vectors come from oracle/, which is pinned by the
reference's analytical tests (tests/test_oracle_kat.py).  Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def tgv3d(N):
    kap = math.pi / N
    return lambda i, x: (-math.sin(kap * x[0]) * math.cos(kap * x[1]) * math.cos(kap * x[2]) if i == 1 else
                         (math.cos(kap * x[0]) * math.sin(kap * x[1]) * math.cos(kap * x[2]) if i == 2 else 0.0))


def ramp_problem(N):
    """Poisson_setup(MultiLevelPoisson, N) of /root/reference/test/test_poisson.jl:1-12 — returns z and the per-iteration log"""
    D = len(N)
    c = np.full(N + (D,), 1.0, dtype=np.float32, order="F")
    orc.BC(c, (0,) * D)
    x, z = np.zeros(N, np.float32, order="F"), np.zeros(N, np.float32, order="F")
    pois = orc.MultiLevelPoisson(x, c, z)
    soln = np.asfortranarray(np.broadcast_to((np.arange(N[0], dtype=np.float32) + 1).reshape((N[0],) + (1,) * (D - 1)), N).copy(order="F"))
    soln -= soln[(1,) * D]
    pois.mult(soln)
    n = pois.solve()
    r1, rinf, w = pois.log()
    return {"n": n, "r1": r1, "rinf": rinf, "omega": w, "x": x.copy(order="F")}


def main():
    orc.build()
    gold = {}
    # (ii) 16³ wall-bounded TGV after 1, 2, 5 mom_step!
    N = 16
    sim = orc.Simulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, u0=tgv3d(N), T=np.float32)
    gold["tgv16_u_init"] = sim.u.copy(order="F")
    for k in range(1, 6):
        sim.step(remeasure=False)
        if k in (1, 2, 5):
            gold[f"tgv16_u_step{k}"] = sim.u.copy(order="F")
            gold[f"tgv16_p_step{k}"] = sim.p.copy(order="F")
    gold["tgv16_dt"] = np.array(sim.dt, dtype=np.float64)
    gold["tgv16_n"] = np.array(sim.pois_n, dtype=np.int32)
    # (iii) 16³ sphere: μ₀/μ₁ from the closed-form measure!, 3 steps, pressure_force
    R = 3.0
    c = (N / 2 - 1,) * 3
    sim = orc.Simulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 3700, body=("sphere", c, R), T=np.float32)
    gold["sph16_mu0"] = sim.field("mu0").copy(order="F")
    gold["sph16_mu1"] = sim.field("mu1").copy(order="F")
    for _ in range(3):
        sim.step(remeasure=False)
    gold["sph16_u_step3"] = sim.u.copy(order="F")
    gold["sph16_force"] = sim.pressure_force()
    gold["sph16_n"] = np.array(sim.pois_n, dtype=np.int32)
    # (iv) the reference's ramp problem: iteration counts and per-iteration norms
    for name, Ns in (("ramp66x66", (66, 66)), ("ramp18c", (18, 18, 18)), ("ramp34c", (34, 34, 34))):
        r = ramp_problem(Ns)
        gold[f"{name}_n"] = np.array([r["n"]], dtype=np.int32)
        gold[f"{name}_r1"], gold[f"{name}_rinf"], gold[f"{name}_omega"] = r["r1"], r["rinf"], r["omega"]
        if name != "ramp34c":
            gold[f"{name}_x"] = r["x"]
    # (i) per-kernel I/O on small random fields (incl. zero coefficients so that iD==0 occurs)
    rng = np.random.default_rng(20261004)
    for tag, shape in (("k2", (10, 10)), ("k3", (10, 9, 8))):
        D = len(shape)
        L = np.asfortranarray(rng.uniform(0, 1, shape + (D,)).astype(np.float32))
        L[L < 0.15] = 0
        L[tuple(slice(2, 4) for _ in range(D))] = 0
        orc.BC(L, (0,) * D)
        x, z = (np.asfortranarray(rng.uniform(-1, 1, shape).astype(np.float32)) for _ in range(2))
        r0 = np.zeros(shape, np.float32, order="F")
        r0[tuple(slice(1, -1) for _ in range(D))] = rng.uniform(-1, 1, tuple(n - 2 for n in shape)).astype(np.float32)
        u = np.asfortranarray(rng.uniform(-1, 1, shape + (D,)).astype(np.float32))
        gold[f"{tag}_L"], gold[f"{tag}_x"], gold[f"{tag}_z"], gold[f"{tag}_r0"], gold[f"{tag}_u"] = L, x, z, r0, u
        po = orc.Poisson(x.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
        gold[f"{tag}_D"], gold[f"{tag}_iD"] = po.field("D").copy(order="F"), po.field("iD").copy(order="F")
        po.field("r")[...] = r0
        po.GaussSeidelRB(it=4, w=0.9)
        gold[f"{tag}_gs_eps"], gold[f"{tag}_gs_r"], gold[f"{tag}_gs_x"] = (po.field(k).copy(order="F") for k in ("eps", "r", "x"))
        po.Jacobi()
        gold[f"{tag}_jac_r"], gold[f"{tag}_jac_x"] = po.field("r").copy(order="F"), po.field("x").copy(order="F")
        rr = np.zeros(shape + (D,), np.float32, order="F")
        Phi = np.zeros(shape, np.float32, order="F")
        orc.conv_diff(rr, u, Phi, nu=0.07)
        gold[f"{tag}_convdiff"] = rr
        ub = u.copy(order="F")
        orc.BC(ub, (1.0, 0.5, -0.25)[:D])
        gold[f"{tag}_bc"] = ub
    np.savez_compressed(os.path.join(OUT, "golden_r01.npz"), **gold)
    print("wrote", os.path.join(OUT, "golden_r01.npz"), f"{os.path.getsize(os.path.join(OUT, 'golden_r01.npz')) / 1024:.0f} KiB,", len(gold), "arrays")


if __name__ == "__main__":
    main()
