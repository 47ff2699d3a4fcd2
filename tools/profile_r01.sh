#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):
#   1. fp64 ceiling probe (v_fma_f64 / v_mfma_f64_16x16x4_f64)
#   2. kernel trace + stats of the default bench command
#   3. PMC passes in SEPARATE runs (MI355X guide: FETCH_SIZE and WRITE_SIZE do not fit one pass; never
#      combine --pmc with sys/hip/hsa tracing): FETCH_SIZE, WRITE_SIZE, SQ busy/wait/VALU/MFMA counters
# Raw output lands in gpurun_out/prof_r01*/ ; tools/summarize_prof.py distils it into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -c "
from x_multi_agent_amd import engine
e = engine.LabEngine(10, 0, 10)   # the probe kernels are in the lab build (include/xk_lab.h)
import json
print(json.dumps({'fp64_mfma_tflops': e.probe_fp64_peak(True), 'fp64_fma_tflops': e.probe_fp64_peak(False)}))
" > $OUT/fp64_peak.json 2>$OUT/fp64_peak.err
cat $OUT/fp64_peak.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r01_trace -o r01 -- python bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r01_fetch -o r01 -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_r01_write -o r01 -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/prof_r01_sq -o r01 -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_sq.log 2>&1
find $OUT -name "*.csv" | wc -l
