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

// v_permlane16_swap_b32 with both operands = the lane's value: out[lane] = {first result, second result}.  Expected: the
// first holds v[lane - 16] in the odd 16-lane rows (own value in the even rows), the second v[lane + 16] in the even rows.
__global__ void k_selftest_permlane16(const unsigned* __restrict__ in, unsigned* __restrict__ out) {
    const unsigned v = in[threadIdx.x];
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[2 * threadIdx.x] = r[0];
    out[2 * threadIdx.x + 1] = r[1];
}

// Does a cross-lane exchange executed by one workgroup disturb ANOTHER workgroup's LDS-DMA on the same CU?  Even workgroups:
// `iters` times { clear a 4 KB LDS tile, LDS-DMA a known 4 KB pattern into it (4 waves x 64 lanes x 16 B), wait, compare }
// and count mismatching dwords.  Odd workgroups keep the CU's other slots busy with `spam`: 0 = plain VALU work, 1 =
// ds_bpermute_b32 (what __shfl_xor(v, 16) compiles to), 2 = v_permlane16_swap_b32, 3 = ds_read / ds_write traffic.
__global__ __launch_bounds__(256) void k_selftest_dma_vs_xlane(const uint4* __restrict__ pattern, int iters, int spam,
                                                               unsigned* __restrict__ errors, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) uint4 T[256];
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    if ((blockIdx.x & 1) == 0) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(pattern), 0, 256 * 16, 0x00020000);
        unsigned bad = 0;
        for (int it = 0; it < iters; ++it) {
            T[l] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
            __syncthreads();
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(T + wave * 64), 16, l * 16, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint4 got = T[l], want = pattern[l];
            bad += (got.x != want.x) + (got.y != want.y) + (got.z != want.z) + (got.w != want.w);
            __syncthreads();
        }
        if (bad) atomicAdd(errors, bad);
    } else {
        unsigned v = l * 2654435761u + blockIdx.x;
        for (int it = 0; it < iters * 8; ++it) {
            if (spam == 1) {
                v = __builtin_amdgcn_ds_bpermute((lane ^ 16) * 4, v) + 1u;
            } else if (spam == 2) {
                const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
                v = (((lane >> 4) & 1) ? r[0] : r[1]) + 1u;
            } else if (spam == 3) {
                T[l].x = v;
                __syncthreads();
                v = T[l ^ 16].x + 1u;
                __syncthreads();
            } else {
                v = v * 1664525u + 1013904223u;
            }
        }
        if (v == 0x12345678u) sink[0] = v;
    }
}

// Do LDS-DMA operations retire in ISSUE ORDER on the vmcnt counter when a younger one has nothing to fetch?  Each wave:
// LDS tile preset to 0xAA; DMA #1 = 1 KB from a cold, wave-private place in `src`; DMA #2 (young) into another tile with
// kind 1 = a ZERO-SIZE descriptor, 2 = an out-of-range voffset on the real descriptor, 0 = a second real load;
// s_waitcnt vmcnt(1) -- "all but the youngest are done" -- then DMA #1's tile is read.  A lane that still sees 0xAA while
// src holds something else is counted: the counted wait let the wave through before the OLDER transfer had landed.
__global__ __launch_bounds__(256) void k_selftest_dma_order(const uint4* __restrict__ src, long src_elems, int kind,
                                                            unsigned* __restrict__ errors) {
    __shared__ __attribute__((aligned(16))) uint4 T[2][256];
    const int l = threadIdx.x, wave = l >> 6;
    T[0][l] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    T[1][l] = T[0][l];
    __syncthreads();
    const long base = (((long)blockIdx.x * 4 + wave) * 7919L * 64) % (src_elems - 64);        // scattered: cold lines
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, (int)(src_elems * 16 < 0x7fffffffL ? src_elems * 16 : 0x7fffffffL), 0x00020000);
    __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, 0, 0x00020000);
    const int lane = l & 63;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(&T[0][wave * 64]), 16, (int)((base + lane) * 16), 0, 0, 0);
    if (kind == 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (__attribute__((address_space(3))) void*)(&T[1][wave * 64]), 16, lane * 16, 0, 0, 0);
    else if (kind == 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(&T[1][wave * 64]), 16, (int)0x80000000, 0, 0, 0);
    else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(&T[1][wave * 64]), 16, (int)((base + lane) * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    const uint4 got = T[0][l];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 want = src[base + lane];
    const bool stale = got.x == 0xAAAAAAAAu && got.y == 0xAAAAAAAAu && (want.x != 0xAAAAAAAAu || want.y != 0xAAAAAAAAu);
    const bool wrong = !stale && (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w);
    if (stale) atomicAdd(errors, 1u);
    if (wrong) atomicAdd(errors + 1, 1u);
}

// The ring protocol of the LDS-DMA conv kernels in miniature, beside another workgroup's cross-lane traffic.  Even workgroups:
// three 4 KB slots; step k: s_waitcnt vmcnt(1) (own piece of step k landed, step k+1's may be in flight), raw s_barrier, issue
// the piece of step k+2 into the slot step k-1 used, then every lane reads 16 bytes of slot k % 3 that ANOTHER wave fetched and
// compares with the source.  Odd workgroups: `spam` as in k_selftest_dma_vs_xlane.
__global__ __launch_bounds__(256) void k_selftest_ring_vs_xlane(const uint4* __restrict__ src, int steps, int spam,
                                                                unsigned* __restrict__ errors, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) uint4 T[3][256];
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    if ((blockIdx.x & 1) == 0) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, steps * 4096, 0x00020000);
        auto issue = [&](int k) {                              // past the end: out of range = zeros, same piece count
            const int vo = k < steps ? (k * 256 + l) * 16 : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(&T[k % 3][wave * 64]), 16, vo, 0, 0, 0);
        };
        issue(0);
        issue(1);
        unsigned bad = 0;
        const int peer = (l * 7 + 67) & 255;                   // a lane of another wave's piece
        for (int k = 0; k < steps; ++k) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(k + 2);
            const uint4 got = T[k % 3][peer];
            const uint4 want = src[k * 256 + peer];
            bad += (got.x != want.x) + (got.y != want.y) + (got.z != want.z) + (got.w != want.w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (bad) atomicAdd(errors, bad);
    } else {
        unsigned v = l * 2654435761u + blockIdx.x;
        for (int it = 0; it < steps * 4; ++it) {
            if (spam == 1) {
                v = __builtin_amdgcn_ds_bpermute((lane ^ 16) * 4, v) + 1u;
            } else if (spam == 2) {
                const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
                v = (((lane >> 4) & 1) ? r[0] : r[1]) + 1u;
            } else if (spam == 3) {
                T[0][l].x = v;
                __syncthreads();
                v = T[0][l ^ 16].x + 1u;
                __syncthreads();
            } else {
                v = v * 1664525u + 1013904223u;
            }
        }
        if (v == 0x12345678u) sink[0] = v;
    }
}
}  // namespace

extern "C" int yolo_selftest_ring_vs_xlane(const void* src, int steps, int blocks, int spam, void* errors, void* sink, hipStream_t st) {
    if (steps < 3 || steps > 100000 || blocks < 2 || spam < 0 || spam > 3) return YOLO_ERR_ARG;
    hipLaunchKernelGGL(k_selftest_ring_vs_xlane, dim3(blocks), dim3(256), 0, st, (const uint4*)src, steps, spam, (unsigned*)errors, (unsigned*)sink);
    return YOLO_LAUNCH_CHECK();
}

// errors[0]: lanes that read the preset after the counted wait (stale); errors[1]: lanes with any other mismatch
extern "C" int yolo_selftest_dma_order(const void* src, long src_bytes, int blocks, int kind, void* errors2, hipStream_t st) {
    if (src_bytes < (1L << 20) || blocks < 1 || kind < 0 || kind > 2) return YOLO_ERR_ARG;
    hipLaunchKernelGGL(k_selftest_dma_order, dim3(blocks), dim3(256), 0, st, (const uint4*)src, src_bytes / 16, kind, (unsigned*)errors2);
    return YOLO_LAUNCH_CHECK();
}

// errors: one zeroed unsigned (mismatching dwords seen by the LDS-DMA workgroups); pattern: 4 KB
extern "C" int yolo_selftest_dma_vs_xlane(const void* pattern4k, int blocks, int iters, int spam, void* errors, void* sink, hipStream_t st) {
    if (blocks < 2 || blocks > 65536 || iters < 1 || spam < 0 || spam > 3) return YOLO_ERR_ARG;
    hipLaunchKernelGGL(k_selftest_dma_vs_xlane, dim3(blocks), dim3(256), 0, st, (const uint4*)pattern4k, iters, spam, (unsigned*)errors,
                       (unsigned*)sink);
    return YOLO_LAUNCH_CHECK();
}

extern "C" int yolo_selftest_permlane16(const void* in64, void* out128, hipStream_t st) {
    hipLaunchKernelGGL(k_selftest_permlane16, dim3(1), dim3(64), 0, st, (const unsigned*)in64, (unsigned*)out128);
    return YOLO_LAUNCH_CHECK();
}

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
