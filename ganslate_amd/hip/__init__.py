"""ctypes binding of libganslate_hip.so (C ABI in include/ganslate_hip.h)."""
