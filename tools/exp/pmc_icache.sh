#!/bin/bash
export XK_LIB_PATH=${XK_LIB_PATH:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/x_multi_agent_amd/lab/libxk.so}   # lab build: env switches, hooks, probes
# instruction-fetch side of the single-launch CAQR: does the unrolled step code of two roles thrash the I-cache two CUs share?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
B="python bench.py --config 4 --no-cpu --no-frame-loop --no-other-configs --steps 5 --warmup 1"
for set in "SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/pmc_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- $B > /dev/null 2>&1
  python - "$OUT/pmc_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xk_caqr_pipe" in r["Kernel_Name"]:
            e = acc[r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
print({k: round(v[0] / max(1, v[1])) for k, v in acc.items()})
PY
done
