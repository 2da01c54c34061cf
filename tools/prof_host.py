import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--model", "DiffMa-L/2", "--batch-per-gpu", "8", "--steps", "30", "--warmup", "5", "--cpu-steps", "0", "--no-extras"]
sys.path.insert(0, ".")
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue()[:9000])
