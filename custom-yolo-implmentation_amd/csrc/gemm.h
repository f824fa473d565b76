// Small batched bf16/f16 MFMA GEMM used by the attention core (and reusable for any token-level matmul):
//   C[b](m,n) (=|+=) alpha * sum_k A[b](m,k) * B[b](k,n)
// Each operand may be stored with K contiguous ("kcontig": A as [M][K], B as [N][K]) or with the other
// index contiguous (A as [K][M], B as [K][N]).  Tiles are staged in LDS in their natural global layout;
// K-contiguous tiles feed the MFMA with ds_read_b128, the others with two ds_read_b64_tr_b16 (the
// hardware transposing read; lane mapping pinned by tests/test_gpu_selftest.py).
#pragma once
#include "common.h"

struct GemmOperand {
    const void* p;
    long b0, b1;      // strides of the two batch indices (batch = b0_idx * nb1 + b1_idx), in elements
    long rs;          // stride between rows of the stored matrix, in elements
    int kcontig;      // 1: K is the contiguous index
};

struct GemmArgs {
    GemmOperand A, B;
    void* C;
    long c_b0, c_b1, c_rs;
    int c_f32;        // 1: C is fp32, 0: C has the operand dtype
    int accumulate;
    float alpha;
    int M, N, K;      // K: loop extent (contiguous-K operands need K % 8 == 0)
    int Kvalid;       // rows k >= Kvalid of a non-kcontig operand read as zero
    int nb0, nb1;
};

int gemm_batched_launch(const GemmArgs& g, int dtype, hipStream_t st);
