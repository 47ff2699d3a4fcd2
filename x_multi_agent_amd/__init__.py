"""x_multi_agent_amd -- MI355X-native xVIO EKF-update engine (libxk.so + bindings).

Only what the hot path needs lives here:
  csrc/       hand-written HIP kernels for gfx950 and the C ABI (include/xk.h)
  engine.py   ctypes binding of the C ABI (tests / bench plumbing)
  synth.py    deterministic synthetic workloads
  fleet.py    one-agent-per-GPU driver, CI payload exchange over RCCL
"""
__all__ = ["engine", "synth"]
