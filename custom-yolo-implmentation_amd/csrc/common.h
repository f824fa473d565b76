// Shared device/host helpers for the gfx950 kernels of libyolo_hip.so.
// Layout convention for every activation tensor: NHWC, element type T in {f32, bf16, f16},
// pixel p = (n*H + h)*W + w, element (p, c) at base[p*ld + c] with ld >= C (ld lets a tensor be a
// channel slice of a wider concat buffer).  Parameters and statistics are fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define YOLO_F32 0
#define YOLO_BF16 1
#define YOLO_F16 2

#define YOLO_OK 0
#define YOLO_ERR_ARG 1001      // unsupported argument combination
#define YOLO_ERR_DTYPE 1002

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

template <typename T> struct vec_of;           // widest 16-byte vector of T
template <> struct vec_of<float>  { static constexpr int N = 4; };
template <> struct vec_of<bf16_t> { static constexpr int N = 8; };
template <> struct vec_of<f16_t>  { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// 16-byte (or narrower) packet of V elements of T
template <typename T, int V> struct alignas(sizeof(T) * V) pack_t { T v[V]; };

template <typename T, int V>
__device__ __forceinline__ void load_pack(const T* p, float (&out)[V]) {
    pack_t<T, V> t = *reinterpret_cast<const pack_t<T, V>*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = to_f<T>(t.v[i]);
}
template <typename T, int V>
__device__ __forceinline__ void store_pack(T* p, const float (&in)[V]) {
    pack_t<T, V> t;
#pragma unroll
    for (int i = 0; i < V; ++i) t.v[i] = from_f<T>(in[i]);
    *reinterpret_cast<pack_t<T, V>*>(p) = t;
}

// split form: issue several raw loads first, convert when the value is consumed (a float array per load
// in flight would double the registers held across the memory latency)
template <typename T, int V>
__device__ __forceinline__ pack_t<T, V> load_raw(const T* p) { return *reinterpret_cast<const pack_t<T, V>*>(p); }
template <typename T, int V>
__device__ __forceinline__ void unpack(const pack_t<T, V>& t, float (&out)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = to_f<T>(t.v[i]);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int hip_status(hipError_t e) { return e == hipSuccess ? YOLO_OK : (int)e; }
#define YOLO_LAUNCH_CHECK() hip_status(hipGetLastError())

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Dynamic LDS above the default limit needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  The attribute
// belongs to the (function, device) pair: `done` is the call site's per-instantiation bitmask of devices that have it,
// set only after a SUCCESSFUL call (a failure is retried by the next launch, not cached; a second device of the same
// process gets its own call).  Benign race: two threads may both set the attribute once.
static inline int yolo_allow_dyn_lds(const void* fn, size_t bytes, unsigned long long& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done, __ATOMIC_RELAXED) & bit) return YOLO_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
    return YOLO_OK;
}

// Zero fill as a KERNEL.  hipMemsetAsync becomes a memset node when the stream is captured into a hipGraph, and on
// this stack replays of such a graph did not always order that node before the kernels that accumulate into the
// buffer (BatchNorm statistics came out wrong from the second replay on, run-dependent).  Kernel nodes of a
// single-stream capture form a plain chain, so everything the training step zeroes goes through here.
static __global__ void k_zero_fill(uint32_t* __restrict__ p, size_t nwords) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline int yolo_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return YOLO_OK;
    if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return hip_status(hipMemsetAsync(p, 0, bytes, st));   // not on the step
    const size_t nwords = bytes / 4;
    size_t blocks = (nwords + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, nwords);
    return YOLO_LAUNCH_CHECK();
}

// dispatch on the activation dtype code
#define YOLO_DISPATCH_T(dtype, ...)                                   \
    switch (dtype) {                                                  \
        case YOLO_F32:  { using T = float;  __VA_ARGS__; } break;     \
        case YOLO_BF16: { using T = bf16_t; __VA_ARGS__; } break;     \
        case YOLO_F16:  { using T = f16_t;  __VA_ARGS__; } break;     \
        default: return YOLO_ERR_DTYPE;                               \
    }
