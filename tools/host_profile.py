"""cProfile of the eager train step's HOST side (autograd's backward thread included via threading.setprofile is not possible for C++ threads: the
backward's Python functions run on autograd's worker thread -- so the step is profiled with torch.autograd.set_multithreading_enabled(False))."""
import cProfile, pstats, sys, io, contextlib
sys.path.insert(0, '.')
import torch
import bench
torch.autograd.set_multithreading_enabled(False)
sys.argv = [sys.argv[0], '--steps', '30', '--warmup', '5', '--no-cpu-baseline', '--primary-only', '--no-roofline', '--launch', 'eager']
pr = cProfile.Profile()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    pr.enable()
    bench.main()
    pr.disable()
print(buf.getvalue().strip().splitlines()[-1][:200])
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
