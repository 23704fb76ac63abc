"""Where does an epoch of the MaxCut demo (d = 16, batch 1) spend its time?"""
import cProfile, pstats, sys, time, io
import numpy as np
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.qobj import Qobj
from diffquantum_amd.sim_plain import SimulatorPlain

p = problems.maxcut_problem()
np.random.seed(0)
sim = SimulatorPlain(lr=2e-2, n_basis=6, n_epoch=5, log_dir="gpurun_out/logs")
sim.T, sim.omegas = p["T"], list(p["omegas"])
args = (Qobj(p["H_cost"]), Qobj(p["H0"]), [Qobj(h) for h in p["Hs"]], Qobj(p["psi0"]))
sim.train_energy(*args)  # warm-up
sim.n_epoch = 60
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
sim.train_energy(*args)
pr.disable()
el = time.time() - t0
print("60 epochs: %.3f s -> %.2f ms/epoch" % (el, el / 60 * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3800])
