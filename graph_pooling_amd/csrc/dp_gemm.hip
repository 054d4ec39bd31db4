// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fma
// chain — bit-for-bit an fmaf chain, so parity with the CPU oracle is reduction-order only).
//
//   C[b] = act(alpha * op(A[b]) * op(B[b]) + beta * C[b] + bias)
//
// This is the general contraction of the DiffPool path: X·W transforms, A·P aggregation
// (encoders.py:965,968), the pooling products S^T Z, S^T A, (S^T A) S (encoders.py:1278-1279) and
// every backward contraction.  Arbitrary M, N, K and leading dimensions (the path's sizes are
// 500, 89, 50, 60 ... — nothing is a multiple of anything), zero-filled edges.
//
// Tiling: 256 threads = 4 waves; each wave owns a 32x32 output block = 2x2 MFMA 16x16 tiles
// (16 accumulator VGPRs); workgroup tile 64x64 (waves 2x2) or 128x32 (waves 4x1) for skinny N.
// K is walked in 32-wide slabs staged through LDS.  The LDS image of an operand keeps the
// operand's own memory orientation so global reads and LDS writes are both lane-contiguous:
//   k-contiguous operand  -> image [r][k], row stride 34  (frag read bank = 2r + k : conflict-free)
//   r-contiguous operand  -> image [k][r], row stride = 16 mod 32 (k rows land on disjoint bank halves)
#include "dp_common.h"

namespace dp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    long sA, sB, sC;
    float alpha, beta;
    int act;
    int tilesN;
};

constexpr int KT = 32;

template <int R, bool KCONTIG>
struct LdsImage {
    // KCONTIG: [R][KT+2]; else [KT][R+16]
    static constexpr int STRIDE = KCONTIG ? (KT + 2) : (R + 16);
    static constexpr int SIZE = KCONTIG ? R * STRIDE : KT * STRIDE;
    static constexpr int PER_THREAD = R * KT / 256;
    __device__ static inline int addr(int r, int k) { return KCONTIG ? r * STRIDE + k : k * STRIDE + r; }
};

// Load one R x KT operand slab (rows r0.., k0..) into registers. `KCONTIG`: element (r,k) at p[r*ld + k],
// else at p[k*ld + r].
template <int R, bool KCONTIG>
__device__ inline void slab_load(const float* __restrict__ p, int ld, int r0, int k0, int rmax, int kmax,
                                 float (&reg)[R * KT / 256]) {
    const int t = threadIdx.x;
    if (KCONTIG) {
        const int k = t & (KT - 1);
        const int rr = t >> 5;  // 0..7
#pragma unroll
        for (int i = 0; i < R * KT / 256; ++i) {
            const int r = rr + i * 8;
            const int gr = r0 + r, gk = k0 + k;
            reg[i] = (gr < rmax && gk < kmax) ? p[(long)gr * ld + gk] : 0.f;
        }
    } else {
        const int r = t & (R - 1);
        const int kk = t / R;  // 0 .. 256/R-1
#pragma unroll
        for (int i = 0; i < R * KT / 256; ++i) {
            const int k = kk + i * (256 / R);
            const int gr = r0 + r, gk = k0 + k;
            reg[i] = (gr < rmax && gk < kmax) ? p[(long)gk * ld + gr] : 0.f;
        }
    }
}

template <int R, bool KCONTIG>
__device__ inline void slab_store(float* lds, const float (&reg)[R * KT / 256]) {
    using L = LdsImage<R, KCONTIG>;
    const int t = threadIdx.x;
    if (KCONTIG) {
        const int k = t & (KT - 1);
        const int rr = t >> 5;
#pragma unroll
        for (int i = 0; i < R * KT / 256; ++i) lds[L::addr(rr + i * 8, k)] = reg[i];
    } else {
        const int r = t & (R - 1);
        const int kk = t / R;
#pragma unroll
        for (int i = 0; i < R * KT / 256; ++i) lds[L::addr(r, kk + i * (256 / R))] = reg[i];
    }
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmArgs a) {
    // A operand: rows = M index, k. Not transposed -> stored [M][K] (k-contiguous).
    using LA = LdsImage<BM, !TA>;
    // B operand: rows = N index, k. Not transposed -> stored [K][N] (r-contiguous); transposed -> [N][K].
    using LB = LdsImage<BN, TB>;
    __shared__ float lds[LA::SIZE + LB::SIZE];
    float* As = lds;
    float* Bs = lds + LA::SIZE;

    const int tile = blockIdx.x;
    const int tm = tile / a.tilesN, tn = tile % a.tilesN;
    const int b = blockIdx.y;
    const float* A = a.A + (long)b * a.sA;
    const float* B = a.B + (long)b * a.sB;
    float* C = a.C + (long)b * a.sC;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WAVES_N = BN / 32;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int l15 = lane & 15, l4 = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float ra[LA::PER_THREAD], rb[LB::PER_THREAD];
    const int nk = (a.K + KT - 1) / KT;
    slab_load<BM, !TA>(A, a.lda, m0, 0, a.M, a.K, ra);
    slab_load<BN, TB>(B, a.ldb, n0, 0, a.N, a.K, rb);
    for (int kt = 0; kt < nk; ++kt) {
        slab_store<BM, !TA>(As, ra);
        slab_store<BN, TB>(Bs, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            slab_load<BM, !TA>(A, a.lda, m0, (kt + 1) * KT, a.M, a.K, ra);
            slab_load<BN, TB>(B, a.ldb, n0, (kt + 1) * KT, a.N, a.K, rb);
        }
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[LA::addr(wr * 32 + i * 16 + l15, kk + l4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[LB::addr(wc * 32 + j * 16 + l15, kk + l4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D map of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 32 + j * 16 + l15;
            if (col >= a.N) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 32 + i * 16 + l4 * 4 + r;
                if (row >= a.M) continue;
                float* cp = C + (long)row * a.ldc + col;
                float v = a.alpha * acc[i][j][r] + bv;
                if (a.beta != 0.f) v += a.beta * (*cp);
                if (a.act == 1) v = fmaxf(v, 0.f);
                *cp = v;
            }
        }
}

template <int BM, int BN>
static void launch_tile(Seq& q, GemmArgs& a, int batch, bool tA, bool tB) {
    const int tilesM = (a.M + BM - 1) / BM;
    a.tilesN = (a.N + BN - 1) / BN;
    dim3 grid(tilesM * a.tilesN, batch), block(256);
    if (!tA && !tB)
        hipLaunchKernelGGL((bgemm_kernel<BM, BN, false, false>), grid, block, 0, q.stream, a);
    else if (!tA && tB)
        hipLaunchKernelGGL((bgemm_kernel<BM, BN, false, true>), grid, block, 0, q.stream, a);
    else if (tA && !tB)
        hipLaunchKernelGGL((bgemm_kernel<BM, BN, true, false>), grid, block, 0, q.stream, a);
    else
        hipLaunchKernelGGL((bgemm_kernel<BM, BN, true, true>), grid, block, 0, q.stream, a);
}

void bgemm(Seq& q, const float* A, const float* B, float* C, const float* bias, int batch, int M, int N,
           int K, int lda, int ldb, int ldc, long sA, long sB, long sC, bool tA, bool tB, float alpha,
           float beta, int act) {
    if (!q.ok() || batch <= 0 || M <= 0 || N <= 0) return;
    if (batch > 65535) {
        set_error("bgemm: batch %d exceeds grid.y limit", batch);
        q.err = DP_ERR_INVALID_ARG;
        return;
    }
    GemmArgs a{A, B, C, bias, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha, beta, act, 0};
    if (N <= 32)
        launch_tile<128, 32>(q, a, batch, tA, tB);
    else
        launch_tile<64, 64>(q, a, batch, tA, tB);
    q.check_launch("bgemm");
}

}  // namespace dp
