"""cProfile of the sharded (multi-GPU) code path with one rank: where does the HOST spend its time per epoch?"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--force-sharded", "--steps", "6", "--warmup", "2", "--cpu-epochs", "0"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
