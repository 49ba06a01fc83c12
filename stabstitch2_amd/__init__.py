"""StabStitch++ inference hot path on MI355X (gfx950): hand-written HIP kernels behind the reference's Python module API.

Importing the package raises the HIP runtime's hardware-queue budget: all HIP streams of a process are multiplexed onto
GPU_MAX_HW_QUEUES AQL queues (default 4), and streams that share a queue execute in order.  The host-fed runners
(pipeline.HostClipRunner / LongVideoStitcher: upload, compute and download streams) and the streaming mode's capture stream
need a queue each or the PCIe copies serialise with the kernels (tools/diag_overlap.py: 9.4 -> 12.1 / 15.3 ms per clip).
The variable is read when the HIP runtime initialises (the first device call of the process -- `import torch` alone does
not), so it only takes effect when this import comes first; an explicit setting by the user wins.
"""
import os as _os

_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
