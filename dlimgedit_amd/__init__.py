"""MI355X-native Segment-Anything path for dlimgedit: ctypes mirror of the C-ABI (api), weight files, sharding helpers."""
import os as _os

# Hardware queues of the HIP runtime (four by default, shared by every stream of the process).  The library's execution
# lanes each want their own queue; the value is read once when the runtime initialises, so it has to be in the
# environment before the first HIP call of the process.  Without it the lanes fall back to stream priorities.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
