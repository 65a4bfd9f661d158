mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_w
for rep in 1 2 3; do
  for v in before varA; do NMI_HIP_LIBRARY=$GRAFT_REPO_ROOT/orbslam2_nmi_amd/lib/libnmi_hip_$v.so python3 bench.py --no-cpu-baseline > gpurun_out/r03_w/${v}_$rep.json 2>>gpurun_out/r03_w/err.log; done
  python3 bench.py --no-cpu-baseline > gpurun_out/r03_w/after_$rep.json 2>>gpurun_out/r03_w/err.log
done
python3 - <<'PY'
import json
for rep in (1,2,3):
    for w in ("before","varA","after"):
        d=json.load(open(f"gpurun_out/r03_w/{w}_{rep}.json"))
        print(rep, w, round(d["value"]/1e6,3), "M evals/s; kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "blocking call ms", round(d["blocking_call_ms"],4))
PY
python3 tools/content_sensitivity.py 2>&1 | tail -8
