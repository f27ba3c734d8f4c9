"""vibravox_amd -- MI355X-native EBEN bandwidth-extension hot path (drop-in for
``vibravox.torch_modules`` / ``vibravox.lightning_modules.eben`` of jhauret/vibravox).

Every op is a hand-written HIP kernel for gfx950 behind the C ABI of ``include/eben_hip.h``;
there is no CPU path (see ``oracle/`` for the test-only CPU restatement).
"""
__version__ = "0.1.0"

from ._env import configure_hw_queues

# data-parallel ranks (torchrun / Lightning DDP export WORLD_SIZE): one hardware queue per stream -- see _env.py; no effect on a
# single-GPU process and none when GPU_MAX_HW_QUEUES is already set
configure_hw_queues()
