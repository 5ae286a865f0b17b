import sys, time, numpy as np
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location("g", "examples/gain_test_headless.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m.run(ticks=50, verbose=False)
t0 = time.perf_counter(); m.run(ticks=500, verbose=False); dt = time.perf_counter() - t0
print(f"headless gain_test: {dt / 500 * 1e6:.0f} us per tick (OSC.generate + state reads + waypoint logic), incl. set-up of the run")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); m.run(ticks=300, verbose=False); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(14)
