"""MI355X-native StrainCall hot path of homopolymer/RAMBL."""
import os as _os

# 16 hardware queues run side by side on one MI355X (more are time-sliced); the library's launch and setup
# streams want one each.  HIP reads this once, when it initialises in the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__all__ = ["cli", "ingest", "samio", "synth", "capi", "stage1", "stage5"]
