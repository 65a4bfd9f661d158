mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_w2
for rep in 1 2 3; do
  for v in before; do NMI_HIP_LIBRARY=$GRAFT_REPO_ROOT/orbslam2_nmi_amd/lib/libnmi_hip_$v.so python3 bench.py --no-cpu-baseline --no-call-site > gpurun_out/r03_w2/${v}_$rep.json 2>>gpurun_out/r03_w2/err.log; done
  python3 bench.py --no-cpu-baseline --no-call-site > gpurun_out/r03_w2/after_$rep.json 2>>gpurun_out/r03_w2/err.log
done
python3 - <<'PY'
import json
for rep in (1,2,3):
    for w in ("before","after"):
        d=json.load(open(f"gpurun_out/r03_w2/{w}_{rep}.json"))
        print(rep, w, round(d["value"]/1e6,3), "M evals/s; kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "blocking call ms", round(d["blocking_call_ms"],4))
PY
timeout -k 10 600 python3 -m pytest tests/test_few_levels.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r03_w2/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_w2/tests.log; tail -3 gpurun_out/r03_w2/tests.log
python3 tools/content_sensitivity.py 2>&1 | tail -8
