"""vibravox_amd -- MI355X-native EBEN bandwidth-extension hot path (drop-in for
``vibravox.torch_modules`` / ``vibravox.lightning_modules.eben`` of jhauret/vibravox).

Every op is a hand-written HIP kernel for gfx950 behind the C ABI of ``include/eben_hip.h``;
there is no CPU path (see ``oracle/`` for the test-only CPU restatement).
"""
__version__ = "0.1.0"

import os as _os

# Data-parallel ranks (torchrun / Lightning DDP export WORLD_SIZE): one hardware queue per stream of the rank -- main, the step's three
# auxiliary streams, the process group's RCCL stream, the graph-capture stream -- instead of the HIP runtime's default four, on which two
# of them share a queue and execute in each other's submission order ([MI355X] single-rank process group: 12.95 -> 12.25 ms/step,
# bench.py --force-ddp).  The runtime reads the variable when it initialises, so this only takes effect when the package is imported
# before the first HIP call; an explicit setting wins.
if int(_os.environ.get("WORLD_SIZE", "1") or "1") > 1:
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
