#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun):  tools/profile_round.sh r02 [config]
#   1. fp64 ceiling probe (v_fma_f64 / v_mfma_f64_16x16x4_f64)
#   2. kernel trace + stats of the default bench command
#   3. PMC passes in SEPARATE runs (MI355X guide: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combine --pmc
#      with sys/hip/hsa tracing): FETCH_SIZE | WRITE_SIZE | SQ busy/wait/VALU/MFMA | L2 hit/miss
# Raw output lands in gpurun_out/prof_<tag>_*/ ; tools/summarize_prof.py <tag> distils it into profiles/ (tracked).
TAG=${1:-r02}
CFG=${2:-4}
SUF=""; [ "$CFG" != "4" ] && SUF="_cfg$CFG"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "$CFG" = "4" ]; then
python -c "
from x_multi_agent_amd import engine
e = engine.LabEngine(10, 0, 10)   # the probe kernels are in the lab build (include/xk_lab.h)
import json
print(json.dumps({'fp64_mfma_tflops': e.probe_fp64_peak(True), 'fp64_fma_tflops': e.probe_fp64_peak(False)}))
" > $OUT/fp64_peak.json 2>$OUT/fp64_peak.err
cat $OUT/fp64_peak.json
fi
B="python bench.py --config $CFG --no-cpu --no-frame-loop --no-other-configs --no-clock-ramp"
rm -rf $OUT/prof_${TAG}${SUF}_*
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}${SUF}_trace -o p -- $B --steps 20 --warmup 3 > $OUT/bench_prof${SUF}.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}${SUF}_fetch -o p -- $B --steps 5 --warmup 1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}${SUF}_write -o p -- $B --steps 5 --warmup 1 > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/prof_${TAG}${SUF}_sq -o p -- $B --steps 5 --warmup 1 > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/prof_${TAG}${SUF}_l2 -o p -- $B --steps 5 --warmup 1 > $OUT/bench_l2.log 2>&1
grep "^{" $OUT/bench_prof${SUF}.log | tail -1 > $OUT/bench_line_${TAG}${SUF}.json
find $OUT/prof_${TAG}${SUF}_* -name "*.csv" | wc -l
