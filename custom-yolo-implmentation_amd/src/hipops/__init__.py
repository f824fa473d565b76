"""ctypes binding of libyolo_hip.so (include/yolo_hip.h) and the autograd Functions built on it."""
