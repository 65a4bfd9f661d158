"""One-off wider campaign of tests/test_render.py::test_gpu_level_cull_by_common_planes_changes_nothing: N more seeds (100..).
    python3 tools/campaign_level_cull.py 40
Round 3: 40 + 20 + 20 seeds, no failure."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("test_render", os.path.join(ROOT, "tests", "test_render.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
bad = []
for seed in range(100, 100 + int(sys.argv[1])):
    try:
        m.test_gpu_level_cull_by_common_planes_changes_nothing(seed)
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        bad.append((seed, tb.lineno, tb.line[:80], str(e)[:80]))
print("CULL CAMPAIGN", sys.argv[1], "seeds, failures:", bad)
