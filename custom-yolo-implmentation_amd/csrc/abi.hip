// ABI stamp: the hash of include/yolo_hip.h's prototypes this library was built against (build.py passes it in).
#include "common.h"
#ifndef YOLO_ABI_HASH
#error "build with custom-yolo-implmentation_amd/build.py (it defines YOLO_ABI_HASH from include/yolo_hip.h)"
#endif
extern "C" long yolo_abi_hash(void) { return YOLO_ABI_HASH; }

// A HIP stream restricted to a subset of the compute units (experiment: the weight-gradient queue on part of the chip so that
// the main chain's kernels always find free CUs).  mask_words 32-bit words, bit i = CU i.  The caller owns the stream
// (torch.cuda.ExternalStream wraps it) and never destroys it before the process ends.
extern "C" int yolo_stream_create_cu_mask(const unsigned int* mask, int mask_words, void** stream_out) {
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask_words, mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)st;
    return YOLO_OK;
}
