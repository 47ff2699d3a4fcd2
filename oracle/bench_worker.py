"""One CPU agent for bench.py's all-core footnote: loads a scenario (.npz), runs `reps` as-written updates with
the C restatement and prints its own compute time.  TEST INFRASTRUCTURE ONLY (see oracle/xk_oracle.c);
only bench.py's cpu_baseline leg starts it.
    python oracle/bench_worker.py scenario.npz liboracle.so reps"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import c_oracle  # noqa: E402

if __name__ == "__main__":
    z = np.load(sys.argv[1])
    sc = {k: (z[k].item() if z[k].shape == () else z[k]) for k in z.files}
    L = c_oracle.lib(sys.argv[2]) if sys.argv[2] != "-" else c_oracle.lib()
    reps = int(sys.argv[3])
    c_oracle.visual_update(sc, library=L)      # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        c_oracle.visual_update(sc, library=L)
    print(f"{time.perf_counter() - t0:.6f}")
