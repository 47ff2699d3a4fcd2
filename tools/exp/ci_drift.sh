export XK_LIB_PATH=${XK_LIB_PATH:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/x_multi_agent_amd/lab/libxk.so}   # lab build: env switches, hooks, probes
for s in 10 50 100 150; do python bench.py --dry-run-ranks 8 --dry-run-rank 1 --steps $s --warmup 20 --no-cpu --no-frame-loop --no-other-configs --dump-posterior /tmp/P_$s.npy > /tmp/o_$s.log 2>&1; python - <<PY
import numpy as np, os
f="/tmp/P_$s.npy"
if os.path.exists(f):
    P=np.load(f); w=np.linalg.eigvalsh(0.5*(P+P.T)); print("steps", $s, "min eig %.3e max %.3e trace %.4e asym %.2e" % (w[0], w[-1], np.trace(P), np.abs(P-P.T).max()))
else:
    print("steps", $s, "failed:", open("/tmp/o_$s.log").read()[-300:])
PY
done
