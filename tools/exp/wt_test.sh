export XK_LIB_PATH=${XK_LIB_PATH:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/x_multi_agent_amd/lab/libxk.so}   # lab build: env switches, hooks, probes
for wt in 0 1; do
  for i in 1 2; do XK_CAQR_WT=$wt python bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('WT=$wt', d['value'], d['ms_per_step'], d['roofline'].get('stages_ms'))"; done
done
