cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3x_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/r3x_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
for v in prev new; do
  lib=$GRAFT_REPO_ROOT/arreau_amd/csrc/libarreau_hip.so; [ $v = prev ] && lib=$GRAFT_REPO_ROOT/arreau_amd/csrc/libarreau_hip_prev.so
  ARREAU_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 60 > gpurun_out/r3x_c2_${v}_$i.json 2> gpurun_out/r3x_c2_${v}_$i.err || { tail -n 20 gpurun_out/r3x_c2_${v}_$i.err; exit 1; }
  ARREAU_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --config c1 --steps 200 > gpurun_out/r3x_c1_${v}_$i.json 2> gpurun_out/r3x_c1_${v}_$i.err || { tail -n 20 gpurun_out/r3x_c1_${v}_$i.err; exit 1; }
done; done
python3 - <<PY
import json
for c in ("c2","c1"):
  for v in ("prev","new"):
    for i in (1,2):
      d=json.load(open("gpurun_out/r3x_%s_%s_%d.json"%(c,v,i)))
      print(c,v,i,"ms_per_step",round(d["ms_per_step"],4),"eager",(d.get("eager_loop") or {}).get("ms_per_step"))
PY
