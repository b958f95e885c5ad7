// f16 MFMA GEMM with fused epilogues:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every dense contraction the reference hands to onnxruntime inside
// Session::run (/root/reference/src/session.cpp:119-136): patch embedding, qkv, proj, fc1, fc2,
// neck convolutions (1x1 and im2col'ed 3x3), and the image-side projections of the mask decoder.
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear [out,in] layout), so
// an MFMA fragment is one 16-byte LDS read.  Structure per workgroup (4 waves, 2x2):
//   * BM x BN output tile, BK = 64; each wave owns a (BM/2) x (BN/2) sub-tile of 32x32 MFMA tiles
//   * operand tiles go HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip), two LDS
//     buffers, tile k+1 in flight while tile k feeds the MFMAs, one barrier per K-tile
//   * LDS image is lane-linear (a DMA wave-instruction writes 8 rows x 128 B); the bank-conflict
//     swizzle chunk ^= (row>>1)&7 is applied on the per-lane SOURCE address and again on the read
//     (cdna_hip_programming.md rule 21), making every ds_read_b128 group conflict-free
//   * XCD-aware workgroup remap so the tiles of one XCD share A panels in its L2
// Epilogue (all optional, fp32): + bias[n], GELU(erf), + residual[m % resid_mod][n], store f32
// and/or f16.
#include "device_common.hpp"
#include "kernels.hpp"

namespace dlimg {
namespace {

constexpr int BK = 64;          // K elements per tile (128 bytes per row)
constexpr int ROW_BYTES = 128;

DLIMG_DEVICE int swz(int row) { return (row >> 1) & 7; }

// Issue the DMA copies of one ROWS x 64 operand tile (rows r0.. of `src`, K offset k0) into `lds`.
template <int ROWS>
DLIMG_DEVICE void stage_tile(const half_t* __restrict__ src, int ld, int r0, int k0, char* lds, int wave, int lane) {
    constexpr int PIECES = ROWS / 8;            // one wave-instruction moves 8 rows x 128 B
#pragma unroll
    for (int q = 0; q < PIECES / 4; ++q) {
        const int p = q * 4 + wave;
        int row = p * 8 + (lane >> 3);
        int chunk = (lane & 7) ^ swz(row);      // source-side swizzle, LDS stays linear
        const half_t* g = src + (size_t)(r0 + row) * ld + k0 + chunk * 8;
        glds16(g, lds + p * 8 * ROW_BYTES);
    }
}

DLIMG_DEVICE half8_t read_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const half8_t*>(lds + row * ROW_BYTES + ((chunk ^ swz(row)) << 4));
}

template <int BM, int BN, int ACT>
__global__ __launch_bounds__(256) void gemm_f16_kernel(k::GemmArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA tiles per wave
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // buffer b: A tile at b*STAGE_BYTES, B tile behind it

    const int lane = lane_id();
    const int wave = wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    const int ntn = a.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;

    float16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

    const int nk = a.K / BK;
    stage_tile<BM>(a.A, a.lda, m0, 0, smem, wave, lane);
    stage_tile<BN>(a.W, a.ldw, n0, 0, smem + A_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tile kt landed; everyone is done with buf cur^1
        if (kt + 1 < nk) {
            char* nxt = smem + (cur ^ 1) * STAGE_BYTES;
            stage_tile<BM>(a.A, a.lda, m0, (kt + 1) * BK, nxt, wave, lane);
            stage_tile<BN>(a.W, a.ldw, n0, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
        }
        const char* la = smem + cur * STAGE_BYTES;
        const char* lb = la + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = read_frag(la, wr * WM + i * 32 + l31, ks * 2 + hi);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = read_frag(lb, wc * WN + j * 32 + l31, ks * 2 + hi);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
        }
    }

    // epilogue
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * WN + j * 32 + l31;
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM + i * 32 + acc_row(r, hi);
                float v = acc[i][j][r] + bias;
                if (ACT == k::ACT_GELU) v = gelu_erf(v);
                if (a.resid) v += a.resid[(size_t)(m % a.resid_mod) * a.ldr + n];
                if (a.out_f32) a.out_f32[(size_t)m * a.ldc32 + n] = v;
                if (a.out_h) a.out_h[(size_t)m * a.ldc16 + n] = (half_t)v;
            }
        }
    }
}

template <int BM, int BN>
void launch(const k::GemmArgs& a, hipStream_t s) {
    const int grid = (a.M / BM) * (a.N / BN);
    const size_t lds = 2 * (BM + BN) * ROW_BYTES;
    if (a.act == k::ACT_GELU)
        hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, k::ACT_GELU>), dim3(grid), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, k::ACT_NONE>), dim3(grid), dim3(256), lds, s, a);
}

}  // namespace

namespace k {

const char* gemm_check(const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return "gemm: empty problem";
    if (a.M % 64 || a.N % 64 || a.K % BK) return "gemm: M, N must be multiples of 64 and K of 64";
    if (a.lda % 8 || a.ldw % 8) return "gemm: operand leading dimensions must be multiples of 8 (16-byte rows)";
    if (a.lda < a.K || a.ldw < a.K) return "gemm: leading dimension smaller than K";
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return "gemm: operands must be 16-byte aligned";
    if (a.resid && a.resid_mod <= 0) return "gemm: resid_mod must be positive";
    if (!a.out_f32 && !a.out_h) return "gemm: no output";
    return nullptr;
}

// Tile choice: the largest tile that still yields at least one workgroup per CU (256), else the
// smallest tile, so a batch-1 encoder GEMM (M = 4096) covers the chip.
void gemm(const GemmArgs& a, hipStream_t s) {
    if (const char* err = gemm_check(a)) throw_error(err);
    auto tiles = [&](int bm, int bn) { return (a.M % bm || a.N % bn) ? 0 : (a.M / bm) * (a.N / bn); };
    if (tiles(128, 128) >= 256) return launch<128, 128>(a, s);
    if (tiles(128, 64) >= 256) return launch<128, 64>(a, s);
    return launch<64, 64>(a, s);
}

}  // namespace k
}  // namespace dlimg
