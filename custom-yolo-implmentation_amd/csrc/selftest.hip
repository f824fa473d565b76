// Hardware self-tests of instruction semantics the tiled kernels rely on (run by tests/test_gpu_selftest.py).
#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS tile T[32 rows][16 cols] of 16-bit values.  Lane l (g = l>>4, i = l&15) issues two
// ds_read_b64_tr_b16 with addresses &T[8g + (i>>2)][4*(i&3)] and &T[8g + 4 + (i>>2)][4*(i&3)].
// Expected (guide T10): out[l][q] = T[8g + q][i], out[l][4+q] = T[8g + 4 + q][i].
__global__ void k_selftest_tr16(const short* __restrict__ in, short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short T[32 * 16];
    const int l = threadIdx.x;
    for (int e = l; e < 32 * 16; e += 64) T[e] = in[e];
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    const short* a1 = &T[(8 * g + (i >> 2)) * 16 + 4 * (i & 3)];
    const short* a2 = &T[(8 * g + 4 + (i >> 2)) * 16 + 4 * (i & 3)];
    s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a2);
#pragma unroll
    for (int q = 0; q < 4; ++q) { out[l * 8 + q] = r1[q]; out[l * 8 + 4 + q] = r2[q]; }
}

// LDS-DMA (buffer_load_dwordx4 ... lds): lane l's 16 bytes land at LDS base + 16*l.  Lanes with (l & 3) == 3 issue
// an OUT-OF-RANGE voffset.  The LDS tile is pre-filled with 0xAA; the test records what those slots hold afterwards
// (zeros = the DMA writes the range-check result, 0xAA = the write is dropped).
__global__ void k_selftest_glds(const uint4* __restrict__ in, uint4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint4 T[64];
    const int l = threadIdx.x;
    T[l] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(in), 0, 128 * 16, 0x00020000);
    const int voff = (l & 3) == 3 ? (int)0x80000000 : ((l * 2) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)T, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[l] = T[l];
}

// `rounds` device-wide barriers among the launch's workgroups (arrival counter per round, relaxed agent-scope polling, one
// acquire fence after; bounded spin: a workgroup that gives up sets *timed_out and leaves).  What a "conv epilogue -> grid
// barrier -> normalise" kernel would pay per layer on top of its work: tools/grid_barrier_cost.py, DESIGN section 6.
template <bool TREE>
__global__ __launch_bounds__(256) void k_selftest_grid_barrier(unsigned* __restrict__ counters, int rounds, int* __restrict__ timed_out,
                                                               float* __restrict__ sink) {
    // TREE: 16 group counters (workgroup index mod 16) whose last arriver adds to the round's top counter -- arrivals no longer
    // serialise on one address; everybody polls the top counter for 16.  Layout per round: [top][16 groups].
    float acc = 0.f;
    const unsigned grp = blockIdx.x & 15u, in_grp = (gridDim.x - grp + 15u) / 16u;
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* top = counters + (TREE ? r * 17 : r);
            unsigned want = gridDim.x;
            if (TREE) {
                want = gridDim.x < 16u ? gridDim.x : 16u;
                const unsigned prev = __hip_atomic_fetch_add(top + 1 + grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1u == in_grp) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { *timed_out = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        acc += (float)r;
    }
    if (sink != nullptr && acc < 0.f) sink[blockIdx.x] = acc;
}
}  // namespace

// counters: zeroed unsigned ints, `rounds` of them (tree 0: one arrival counter per round) or 17 * rounds (tree 1: a top
// counter + 16 group counters per round); blocks must all be resident at once (<= 8 per CU for this kernel)
extern "C" int yolo_selftest_grid_barrier(void* counters, int blocks, int rounds, int tree, int* timed_out, hipStream_t st) {
    if (blocks < 1 || blocks > 2048 || rounds < 0 || rounds > 64) return YOLO_ERR_ARG;
    if (tree)
        hipLaunchKernelGGL(k_selftest_grid_barrier<true>, dim3(blocks), dim3(256), 0, st, (unsigned*)counters, rounds, timed_out, (float*)nullptr);
    else
        hipLaunchKernelGGL(k_selftest_grid_barrier<false>, dim3(blocks), dim3(256), 0, st, (unsigned*)counters, rounds, timed_out, (float*)nullptr);
    return YOLO_LAUNCH_CHECK();
}

extern "C" int yolo_selftest_glds(const void* in128x16, void* out64x16, hipStream_t st) {
    hipLaunchKernelGGL(k_selftest_glds, dim3(1), dim3(64), 0, st, (const uint4*)in128x16, (uint4*)out64x16);
    return YOLO_LAUNCH_CHECK();
}

extern "C" int yolo_selftest_tr16(const void* tile_in, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_selftest_tr16, dim3(1), dim3(64), 0, st, (const short*)tile_in, (short*)out);
    return YOLO_LAUNCH_CHECK();
}
