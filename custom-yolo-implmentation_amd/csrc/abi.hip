// ABI stamp: the hash of include/yolo_hip.h's prototypes this library was built against (build.py passes it in).
#include "common.h"
#ifndef YOLO_ABI_HASH
#error "build with custom-yolo-implmentation_amd/build.py (it defines YOLO_ABI_HASH from include/yolo_hip.h)"
#endif
extern "C" long yolo_abi_hash(void) { return YOLO_ABI_HASH; }
