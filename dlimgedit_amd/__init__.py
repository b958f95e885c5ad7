"""MI355X-native Segment-Anything path for dlimgedit: ctypes mirror of the C-ABI (api), weight files, sharding helpers."""
