import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import waterlily_jl_amd as w
N = 256
sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv_periodic", perdir=(1, 2, 3))
for _ in range(3): sim.mom_step_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): sim.mom_step_()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("periodic TGV 256^3: ms/step", el / 10 * 1e3, "pois.n", sim.pois_n[-6:], "dt", float(sim.dt[-1]))
