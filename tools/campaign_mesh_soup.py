"""One-off wider campaign of tests/test_render.py::test_gpu_mesh_random_soup_vs_twin: N more seeds (or the seeds given).
    python3 tools/campaign_mesh_soup.py 60        # seeds 4..63
    python3 tools/campaign_mesh_soup.py 0 7 23     # just these
A seed whose scene covers the whole image fails the test's own precondition (line 'covered and background both present'): not a parity failure.
Round 3: 60 seeds, 57 valid, all equal to the twin."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("test_render", os.path.join(ROOT, "tests", "test_render.py"))
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
