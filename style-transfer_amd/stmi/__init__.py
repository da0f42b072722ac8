"""MI355X-native engine of the style-transfer hot path: ctypes binding of the
HIP kernel library plus the host-side executors built on it."""
