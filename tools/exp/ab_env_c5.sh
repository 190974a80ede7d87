#!/bin/bash
# usage (GPU box): tools/exp/ab_env_c5.sh "VAR=value ..." ["VAR2=value ..."] -- the training step (bench.py --config c5) with the default
# environment and with each given one, alternating runs in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  i=0
  for envs in "" "$@"; do
    python3 - "$envs" $rep $i <<'PY'
import json, os, subprocess, sys
envs, rep, i = sys.argv[1], sys.argv[2], sys.argv[3]
env = dict(os.environ)
for kv in envs.split():
    k, v = kv.split("=", 1); env[k] = v
out = subprocess.run([sys.executable, "bench.py", "--config", "c5", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
if out.returncode:
    print(out.stderr[-800:]); sys.exit(1)
d = json.loads(out.stdout.strip().splitlines()[-1])
print("rep %s  %-40s step %.4f ms  forward+backward %.4f ms" % (rep, envs or "(default)", d["ms_per_step"], d["forward_backward_ms"]))
PY
    [ $? -eq 0 ] || exit 1
    i=$((i+1))
  done
done
