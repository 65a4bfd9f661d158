import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("test_render", os.path.join(sys.path[0], "test_render.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
t0 = time.time(); bad = []
for seed in (list(map(int, sys.argv[2:])) or range(4, 4 + int(sys.argv[1]))):
    try:
        m.test_gpu_mesh_random_soup_vs_twin(seed)
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        bad.append((seed, tb.lineno, tb.line[:90], str(e)[:100]))
    if (seed % 5) == 0: print("seed", seed, "elapsed %.0f s" % (time.time() - t0), "bad", bad, flush=True)
print("SOUP CAMPAIGN", int(sys.argv[1]), "seeds, failures:", bad)
