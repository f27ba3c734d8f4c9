"""Process-level HIP runtime settings of a data-parallel rank -- the ONE place that decides them.

The HIP runtime multiplexes a process's streams onto ``GPU_MAX_HW_QUEUES`` hardware queues (default 4); streams that share a queue execute
in each other's submission order.  A data-parallel rank of the EBEN step has six streams -- main, the step's three auxiliary streams
(``ops.aux_stream``), the process group's RCCL stream, the graph-capture stream -- so two of them collide at the default: [MI355X]
single-rank process group 12.95 ms/step at 4 queues, 13.5 at 5, **12.25 at 6**, 15.9 at 8; the plain single-GPU step (four streams) is
12.0 at any of them and is left alone.  The runtime reads the variable when it initialises, i.e. this must run before the first HIP
call of the process: ``run.py``, ``bench.py`` and the package import (when the launcher has exported ``WORLD_SIZE > 1``) call it first
thing; an explicit setting in the environment always wins.
"""
import os
import sys

DATA_PARALLEL_HW_QUEUES = "6"


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
