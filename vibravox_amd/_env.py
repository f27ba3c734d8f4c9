"""Process-level HIP runtime settings of a data-parallel rank -- the ONE place that decides them.

The HIP runtime multiplexes a process's streams onto ``GPU_MAX_HW_QUEUES`` hardware queues (default 4); streams that share a queue execute
in each other's submission order.  A data-parallel rank of the EBEN step has six hot streams -- main, the step's three auxiliary streams
(``ops.aux_stream``), the process group's RCCL stream, the graph-capture stream -- so two of them collide at the default: [MI355X]
single-rank process group 12.95 ms/step at 4 queues, 13.5 at 5, **12.25 at 6**, 15.9 at 8 (round 3); the plain single-GPU step (four
streams) is the same at any of them and is left alone.  Round 6 (two-pass backward: more work beside the generator backward; every
setting repeated three times, deterministic to 0.05 ms): 2 queues 11.9, 3 10.5, 4 11.7, 5 10.35, 6 10.9, **7 9.9**, 8 11.9, 9-16 13.0-13.4
against 8.7 ms for the plain step -- which streams end up sharing a queue is the runtime's round-robin over every stream the process
ever created (torch's pools included), so the best count moves with the code; 7 is this tree's.  (The round-5 tree measured on the same
boxes the same day: 10.96-10.99 ms at its 6 queues, where its committed line of the day before says 9.49 -- the pool's runtime changed
under it; all figures here are same-day.)  The runtime reads the variable when it initialises, i.e. this must run before the first HIP
call of the process: ``run.py``, ``bench.py`` and the package import (when the launcher has exported ``WORLD_SIZE > 1``) call it first
thing; an explicit setting in the environment always wins.
"""
import os
import sys

DATA_PARALLEL_HW_QUEUES = "7"


def configure_hw_queues(data_parallel=None) -> str:
    """Sets ``GPU_MAX_HW_QUEUES`` for a data-parallel rank unless the environment already names a value; returns the value in force
    ("" = the runtime's default).  ``data_parallel=None``: decided from ``WORLD_SIZE``."""
    if data_parallel is None:
        data_parallel = int(os.environ.get("WORLD_SIZE", "1") or "1") > 1
    if data_parallel and "GPU_MAX_HW_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = DATA_PARALLEL_HW_QUEUES
        if os.environ.get("EBEN_VERBOSE"):
            print(f"[vibravox_amd] GPU_MAX_HW_QUEUES={DATA_PARALLEL_HW_QUEUES} (data-parallel rank: one hardware queue per stream)", file=sys.stderr)
    return os.environ.get("GPU_MAX_HW_QUEUES", "")
