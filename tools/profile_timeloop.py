"""cProfile of the drop-in time_loop at config-2 step size (run on the GPU box)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ludvm_amd import LUDVM, Engine
eng = Engine(0)
kw = dict(t0=0, tf=float(sys.argv[1]) if len(sys.argv) > 1 else 3.0, dt=1e-3, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30,
          LESPcrit=0.2, Naca="0012", verbose=False, engine=eng, precision="f32", history="sparse")
LUDVM(**dict(kw, tf=0.2))
pr = cProfile.Profile()
pr.enable()
sim = LUDVM(**kw)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
