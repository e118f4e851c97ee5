// Two pointwise convolutions in ONE launch (round 6; the "c3 -> next c1" pair of a bottleneck block, src/ops/conv.rs:248-284 twice):
//
//     y1 = act1(W1 . x + b1 [+ residual])          M1 x N   (K1 = input channels; the block's expand layer)
//     y2 = act2(W2 . y1 + b2)                      M2 x N   (K2 = M1;            the next block's reduce layer)
//
// y1 is still written (the next block's residual Add and possibly a downsample layer read it), but the second convolution takes it from LDS:
// the stage-0 expand layers of ResNet-50 are bound by HBM bytes (231 MB at batch 32: 47 us of a 2.4 ms step, three times), and the reduce
// layer behind each reads the 103 MB tensor straight back.
//
// One workgroup owns BN output columns (pixels) and ALL output rows of both layers.  The first layer is computed in chunks of 64 output
// channels; a finished chunk (bias, residual, activation applied: exactly the value that goes to memory) is parked in LDS as rows
// [64 mc, 64 mc + 64) of the second layer's B operand and multiplied into the second layer's accumulators at once.  The second layer's
// depth index therefore runs 0, 1, 2, ... K2 - 1 in order on one accumulator, k-pair by k-pair, as in igemm_f32_dma_kernel's MODE 0 --
// K1 <= 256 and K2 = M1 <= 256 (one depth block each, rten-gemm/src/lib.rs:1008-1013), so both results are bit-identical to the two
// separate launches.
//
// The first layer's B operand (x: the same for every chunk) lives in registers as MFMA fragments.  LDS (one array): W1s [K1][64] (chunk mc + 1 arrives under
// the second layer's MFMAs of chunk mc) | W2s [64][M2] (arrives under the first layer's) | Ts [64][BN].  K1 = 64, BN = 64, M2 = 64: 48 KB = three workgroups
// per compute unit.
// All tiles arrive by LDS-DMA; the only ordinary loads are the residual / bias values of the NEXT chunk, requested a whole phase early.
#include "gemm_f32_common.h"

namespace {

struct PairArgs {
    const float *X, *W1, *B1, *R, *W2, *B2;
    float *Y1, *Y2;
    int M1, K1, M2, N; // N = images x pixels
    int Pn;            // pixels per image (the row stride of x, y1, y2)
    long long x_ns, y1_ns, y2_ns; // image strides (floats)
    int w1_cs, w2_cs;  // row length of the k-major packed weights (output channels rounded up to 4)
    unsigned x_bytes, w1_bytes, w2_bytes;
    int act1, act2;
    int dbg;     // -DRTEN_ABLATE tuning builds: 1 = no residual, 2 = no y1 store, 4 / 8 = no first- / second-layer MFMAs, 16 = weights loaded once
    // the shortcut form (DS): the residual is itself a pointwise convolution of another tensor -- a stage's first block, whose shortcut 64 -> M1 layer is computed HERE,
    // chunk by chunk, instead of being written by its own launch and read back
    const float *Xd, *Wd, *Bd;
    long long xd_ns;
    int wd_cs;
    unsigned xd_bytes, wd_bytes;
    int pad_[18]; // (the block spans 5 cache lines: kernarg_prefetch takes 3, 5 or 7)
};
static_assert(sizeof(PairArgs) > 256 && sizeof(PairArgs) <= 320, "PairArgs: five cache lines");

typedef __attribute__((address_space(3))) void *lds_ptr_t;
#ifdef RTEN_ABLATE
#define PAIR_DBG(p) ((p).dbg)
#else
#define PAIR_DBG(p) 0
#endif

// acc[i][j] += A[k][m] . B[k][n] over k-pairs [0, NKK): A image [k][lda] at As (lane base folded in), B image [k][ldb].  Operand fragments
// double buffered across k-pairs (the form of igemm_f32_dma_kernel's compute_tile).
template <int TM, int TN, int NKK>
__device__ __forceinline__ void mma_block(const float *As, int lda, const float *Bs, int ldb, f32x16 (&acc)[TM][TN]) {
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) af[0][i] = As[i * 32];
#pragma unroll
    for (int j = 0; j < TN; j++) bf[0][j] = Bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < NKK) {
#pragma unroll
            for (int i = 0; i < TM; i++) af[nxt][i] = As[2 * (kk + 1) * lda + i * 32];
#pragma unroll
            for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[2 * (kk + 1) * ldb + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_iglp_opt(0);
}

template <int BN, int M2, int K1, bool DS = false>
__global__ __launch_bounds__(256, ((K1 + (DS ? 64 : 0)) * 64 + 64 * M2 + 64 * BN) * 4 <= 53 * 1024 ? 3 : (((K1 + (DS ? 64 : 0)) * 64 + 64 * M2 + 64 * BN) * 4 <= 80 * 1024 ? 2 : 1)) void conv_pair_f32_kernel(const PairArgs p) {
    kernarg_prefetch<(int)sizeof(PairArgs)>();
    constexpr int TN = BN / 64, TM2 = M2 / 64;
    constexpr int W1S = 0, WDS = W1S + K1 * 64, W2S = WDS + (DS ? 64 * 64 : 0), TS = W2S + 64 * M2, TOTAL = TS + 64 * BN; // (the shortcut's depth is 64)
    __shared__ __attribute__((aligned(16))) float smem[TOTAL];
    constexpr int NW1 = K1 * 64 / 1024, NW2 = 64 * M2 / 1024; // dwordx4 DMA instructions per wave
    static_assert(NW1 >= 1 && NW2 >= 1, "tiles too small for the four-wave DMA split");

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6, wm = wq >> 1, wn = wq & 1;

    int tile;
    {   // workgroups of one XCD take a contiguous range of column tiles (as the GEMM kernels do)
        const int id = blockIdx.x, nt = (int)gridDim.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int n0 = tile * BN;
    const int nch = p.M1 >> 6;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.W1, 0, (int)p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.W2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)(p.R ? p.R : p.Y1), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y1, 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y2, 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B1 ? p.B1 : p.W1), 0, p.B1 ? p.M1 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B2 ? p.B2 : p.W2), 0, p.B2 ? p.M2 * 4 : 0, 0x00020000);

    auto issue_w1 = [&](int mc) { // rows [0, K1) x output channels [64 mc, 64 mc + 64) of the k-major weights -> W1s[mc & 1]
        float *dst = smem + W1S;
        const unsigned soff = (unsigned)mc * 256u;
#pragma unroll
        for (int j = 0; j < NW1; j++) {
            const int f = (wave * NW1 + j) * 256 + lane * 4;
            const unsigned voff = mc < nch ? (unsigned)(((f >> 6) * p.w1_cs + (f & 63)) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr_t)(dst + (wave * NW1 + j) * 256), 16, (int)voff, (int)soff, 0, 0);
        }
    };
    auto issue_w2 = [&](int mc) { // depth rows [64 mc, 64 mc + 64) x all M2 output channels of the second layer's weights
        const unsigned soff = (unsigned)(mc * 64 * p.w2_cs) * 4u;
#pragma unroll
        for (int j = 0; j < NW2; j++) {
            const int f = (wave * NW2 + j) * 256 + lane * 4;
            const unsigned voff = (unsigned)(((f / M2) * p.w2_cs + (f % M2)) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (lds_ptr_t)(smem + W2S + (wave * NW2 + j) * 256), 16, (int)voff, (int)soff, 0, 0);
        }
    };
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsWd = __builtin_amdgcn_make_buffer_rsrc((void *)(DS ? p.Wd : p.W1), 0, DS ? (int)p.wd_bytes : 0, 0x00020000);
    [[maybe_unused]] auto issue_wd = [&](int mc) { // the shortcut's weights: depth rows [0, 64) x output channels [64 mc, 64 mc + 64)
        const unsigned soff = (unsigned)mc * 256u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = (wave * 4 + j) * 256 + lane * 4;
            const unsigned voff = mc < nch ? (unsigned)(((f >> 6) * p.wd_cs + (f & 63)) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWd, (lds_ptr_t)(smem + WDS + (wave * 4 + j) * 256), 16, (int)voff, (int)soff, 0, 0);
        }
    };
    issue_w1(0);
    if constexpr (DS) issue_wd(0);

    // ---- this lane's output columns: offsets into y1 / residual and y2 (column part; the row rides in the scalar offset)
    unsigned col1[TN], col2[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        const bool ok = n < p.N;
        const int nn = ok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        // (the lane's first row -- its wave's row block and the 4-row shift of lanes 32..63 -- is part of the lane offset: the scalar offset must be wave-uniform)
        col1[j] = ok ? (unsigned)(((long long)nb * p.y1_ns + np + (long long)(wm * 32 + 4 * half) * p.Pn) * 4) : OOB;
        col2[j] = ok ? (unsigned)(((long long)nb * p.y2_ns + np + (long long)(wm * (M2 / 2) + 4 * half) * p.Pn) * 4) : OOB;
    }
    const unsigned rs4 = (unsigned)p.Pn << 2;
    unsigned colw = OOB; // y1 stores: thread t moves 4 pixels of row t / (BN / 4) (+ 1024 / BN rows per instruction) of a chunk
    {
        const int n = n0 + (t % (BN / 4)) * 4;
        if (n < p.N) {
            const int nb = n / p.Pn, np = n - nb * p.Pn;
            colw = (unsigned)(((long long)nb * p.y1_ns + np + (long long)(t / (BN / 4)) * p.Pn) * 4);
        }
    }
    const unsigned brow1 = (unsigned)(wm * 32 + 4 * half) << 2, brow2 = (unsigned)(wm * (M2 / 2) + 4 * half) << 2;

    // ---- the first layer's B operand (x, [K1][this wave's columns]) is the same for every chunk: it lives in REGISTERS as MFMA fragments (lane (l31, half) holds
    // x[2 kk + half][column l31] for every k-pair kk), loaded straight from memory -- no LDS image, no ds_read for it in the chunk loop
    float xb[K1 / 2][TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        const bool ok = n < p.N;
        const int nn = ok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const unsigned colx = ok ? (unsigned)(((long long)nb * p.x_ns + np + (long long)half * p.Pn) * 4) : OOB;
#pragma unroll
        for (int kk = 0; kk < K1 / 2; kk++) xb[kk][j] = buf_load1(rsX, colx, (unsigned)(2 * kk) * rs4);
    }
    // (DS: the same for the shortcut's operand -- another tensor of 64 channels over the same pixels)
    [[maybe_unused]] float xdb[DS ? 32 : 1][TN];
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsXd = __builtin_amdgcn_make_buffer_rsrc((void *)(DS ? p.Xd : p.X), 0, DS ? (int)p.xd_bytes : 0, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsBd = __builtin_amdgcn_make_buffer_rsrc((void *)((DS && p.Bd) ? p.Bd : p.W1), 0, (DS && p.Bd) ? p.M1 * 4 : 0, 0x00020000);
    if constexpr (DS) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * (BN / 2) + j * 32 + l31;
            const bool ok = n < p.N;
            const int nn = ok ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            const unsigned colx = ok ? (unsigned)(((long long)nb * p.xd_ns + np + (long long)half * p.Pn) * 4) : OOB;
#pragma unroll
            for (int kk = 0; kk < 32; kk++) xdb[kk][j] = buf_load1(rsXd, colx, (unsigned)(2 * kk) * rs4);
        }
    }
    const bool has_res = !DS && p.R != nullptr && !(PAIR_DBG(p) & 1);

    // residual and bias of one chunk: requested a phase before their use
    float rr[TN][16], b1[16];
    [[maybe_unused]] float bd[16];
    auto fetch_res = [&](int mc) {
        const unsigned row0 = (unsigned)(mc * 64); // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; r++) b1[r] = buf_load1(rsB1, brow1, (row0 + (unsigned)acc_row(r)) << 2);
        if constexpr (DS) { // the shortcut's bias rows of this chunk take the place of the residual requests (rr itself is computed, below)
#pragma unroll
            for (int r = 0; r < 16; r++) bd[r] = buf_load1(rsBd, brow1, (row0 + (unsigned)acc_row(r)) << 2);
        } else if (has_res) {
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) rr[j][r] = buf_load1(rsR, col1[j], (row0 + (unsigned)acc_row(r)) * rs4);
        }
    };
    fetch_res(0);

    f32x16 acc2[TM2][TN];
#pragma unroll
    for (int i = 0; i < TM2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[i][j][r] = 0.f;

    const float *a1 = smem + W1S + half * 64 + wm * 32 + l31;
    [[maybe_unused]] const float *ad = smem + WDS + half * 64 + wm * 32 + l31;
    const float *a2 = smem + W2S + half * M2 + wm * (M2 / 2) + l31;
    const float *bt = smem + TS + half * BN + wn * (BN / 2) + l31;
    float *tw = smem + TS + (wm * 32 + 4 * half) * BN + wn * (BN / 2) + l31;

    wait_vmcnt<0>(); // x, the first weight chunk (and the first residual rows)
    __builtin_amdgcn_s_barrier();

    // Per chunk (one W1 buffer; every tile arrives a phase before its use, the waits below name what they wait for):
    //   [E]  issue W2[mc] | first-layer MFMAs | wait: W2[mc], residual rows mc | bias, residual, activation -> Ts
    //   [B]  everyone is past the first-layer MFMAs: issue W1[mc + 1]; THEN this chunk's stores and the next chunk's residual requests (younger than the DMA:
    //        the counted wait below leaves exactly those in flight) | second-layer MFMAs | wait: W1[mc + 1]
    constexpr int NSTORE = 64 * BN / 1024; // 16-byte store instructions per chunk per lane
    constexpr int YOUNGER = 16 + 16 * TN + NSTORE < 63 ? 16 + 16 * TN + NSTORE : 63; // bias + residual requests + stores issued after the W1 DMA (vmcnt is 6 bits)
    for (int mc = 0; mc < nch; mc++) {
        if (!(PAIR_DBG(p) & 16) || mc == 0) issue_w2(mc);

        f32x16 acc1[1][TN];
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[0][j][r] = 0.f;
        if (!(PAIR_DBG(p) & 4)) { // A fragments from W1s -- all of the chunk's requested up front: one accumulator per wave, so nothing should sit between its MFMAs --, B fragments from registers
            float afa[K1 / 2];
#pragma unroll
            for (int kk = 0; kk < K1 / 2; kk++) afa[kk] = a1[2 * kk * 64];
            __builtin_amdgcn_sched_barrier(0); // every ds_read of the chunk is issued before its first MFMA
#pragma unroll
            for (int kk = 0; kk < K1 / 2; kk++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc1[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afa[kk], xb[kk][j], acc1[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (DS) { // the shortcut's chunk: Wd[:, 64 mc ..] . xd on an accumulator of its own, then + its bias = the value the separate launch would have written
            f32x16 accd[TN];
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) accd[j][r] = 0.f;
            float afd[32];
#pragma unroll
            for (int kk = 0; kk < 32; kk++) afd[kk] = ad[2 * kk * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 32; kk++)
#pragma unroll
                for (int j = 0; j < TN; j++) accd[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afd[kk], xdb[kk][j], accd[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<0>(); // (the bias rows)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) rr[j][r] = p.Bd ? accd[j][r] + bd[r] : accd[j][r];
        }
        wait_vmcnt<0>(); // W2 chunk mc (requested before the MFMAs above), this chunk's residual rows and the previous chunk's stores (a whole phase old)
        const unsigned row0 = (unsigned)(mc * 64);
        f32x16 v[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            v[j] = acc1[0][j];
            if (p.B1) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[j][r] = v[j][r] + b1[r];
            }
            if (DS || has_res) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[j][r] = v[j][r] + rr[j][r];
            }
            if (p.act1 == RTEN_HIP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[j][r] = vm::relu(v[j][r]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) tw[acc_row(r) * BN + j * 32] = v[j][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // [B] Ts complete, W2s landed for everyone; nobody reads W1s any more

        if (!(PAIR_DBG(p) & 16)) issue_w1(mc + 1);
        if constexpr (DS) issue_wd(mc + 1);
        asm volatile("" ::: "memory"); // (program order: the DMA above is older than everything below)
        if (!(PAIR_DBG(p) & 2)) {
            // the finished chunk goes to memory from its LDS image: 16 bytes per lane, a row of BN pixels = BN * 4 contiguous bytes
#pragma unroll
            for (int q = 0; q < NSTORE; q++) {
                const f32x4 o = *(const f32x4 *)(smem + TS + q * 1024 + t * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsY1, (int)colw, (int)((row0 + (unsigned)(q * (1024 / BN))) * rs4), 0);
            }
        }
        if (mc + 1 < nch) fetch_res(mc + 1);
        asm volatile("" ::: "memory");

        if (!(PAIR_DBG(p) & 8)) mma_block<TM2, TN, 32>(a2, M2, bt, BN, acc2);
        // W1 chunk mc + 1 has landed: exactly the stores / requests issued after its DMA may still be in flight (without a residual there are 16 * TN fewer of them)
        if constexpr (DS) wait_vmcnt<(16 + 16 + NSTORE)>(); // (two bias requests of 16 rows, the stores)
        else if (has_res) wait_vmcnt<YOUNGER>();
        else wait_vmcnt<YOUNGER - 16 * TN>();
        __builtin_amdgcn_s_barrier(); // [E] ... for everyone; W2s and Ts are free
    }

    // ---- second layer's epilogue: bias, activation, store
#pragma unroll
    for (int i = 0; i < TM2; i++) {
        const unsigned row0 = (unsigned)(i * 32);
        float b2[16];
#pragma unroll
        for (int r = 0; r < 16; r++) b2[r] = buf_load1(rsB2, brow2, (row0 + (unsigned)acc_row(r)) << 2);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            f32x16 v = acc2[i][j];
            if (p.B2) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] + b2[r];
            }
            if (p.act2 == RTEN_HIP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = vm::relu(v[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v[r]), rsY2, (int)col2[j], (int)((row0 + (unsigned)acc_row(r)) * rs4), 0);
        }
    }
}

bool pointwise_unit(const rten_hip_conv2d_desc *d) {
    return d->kh == 1 && d->kw == 1 && d->groups == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pads[0] == 0 && d->pads[1] == 0 && d->pads[2] == 0 && d->pads[3] == 0;
}

} // namespace

// 1 = this pair of descriptors has a one-launch form (rten_hip_conv2d_f32_pair would accept it), 0 = it has none.
RTEN_EXPORT int32_t rten_hip_conv2d_f32_pair_supported(const rten_hip_conv2d_desc *d1, const rten_hip_conv2d_desc *d2) {
    if (!d1 || !d2) return 0;
    if (!pointwise_unit(d1) || !pointwise_unit(d2)) return 0;
    if (d2->c != d1->o || d2->n != d1->n || d2->h != d1->out_h || d2->w != d1->out_w) return 0;
    if (d1->c != 64 || d1->o % 64 || d1->o > 256 || d1->o < 64) return 0; // K1 = 64 (ResNet's stage 0), K2 = M1 <= 256: one depth block each
    if (d2->o != 64 && d2->o != 128) return 0;
    const long long P = (long long)d1->out_h * d1->out_w;
    if (P % 4 || P <= 0 || d1->n <= 0) return 0;
    if ((long long)d1->n * d1->o * P * 4 > 0x7fffffffLL || (long long)d1->n * d1->c * P * 4 > 0x7fffffffLL || (long long)d2->n * d2->o * P * 4 > 0x7fffffffLL) return 0; // 32-bit buffer offsets
    return 1;
}

RTEN_EXPORT int32_t rten_hip_conv2d_f32_pair(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d1, const float *x, const float *w1_packed, const float *bias1,
                                             const float *residual, uint32_t flags1, float *y1, const rten_hip_conv2d_desc *d2, const float *w2_packed,
                                             const float *bias2, uint32_t flags2, float *y2) {
    RTEN_CHECK_CTX(ctx);
    if (!d1 || !d2 || !x || !w1_packed || !w2_packed || !y1 || !y2) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv pair: NULL operand");
    if ((flags1 & RTEN_HIP_CONV_RESIDUAL) && !residual) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv pair: residual flag without residual tensor");
    if (flags2 & RTEN_HIP_CONV_RESIDUAL) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair: the second convolution takes no residual");
    if (!rten_hip_conv2d_f32_pair_supported(d1, d2)) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair: no one-launch form for these two convolutions (rten_hip_conv2d_f32_pair_supported)");
    if (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)w2_packed) & 15) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair: operands must be 16-byte aligned");
    const int P = d1->out_h * d1->out_w;
    PairArgs a = {};
    a.X = x; a.W1 = w1_packed; a.B1 = bias1; a.R = (flags1 & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr; a.W2 = w2_packed; a.B2 = bias2;
    a.Y1 = y1; a.Y2 = y2;
    a.M1 = d1->o; a.K1 = d1->c; a.M2 = d2->o; a.N = d1->n * P; a.Pn = P;
    a.x_ns = (long long)d1->c * P; a.y1_ns = (long long)d1->o * P; a.y2_ns = (long long)d2->o * P;
    a.w1_cs = (d1->o + 3) & ~3; a.w2_cs = (d2->o + 3) & ~3;
    a.x_bytes = (unsigned)((long long)d1->n * d1->c * P * 4);
    a.w1_bytes = (unsigned)((long long)d1->c * a.w1_cs * 4);
    a.w2_bytes = (unsigned)((long long)d2->c * a.w2_cs * 4);
    a.act1 = (flags1 & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;
    a.act2 = (flags2 & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;
    a.dbg = ctx->debug >> 8;
    const int tiles = (a.N + 63) / 64;
    if (a.M2 == 64) hipLaunchKernelGGL((conv_pair_f32_kernel<64, 64, 64>), dim3(tiles), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv_pair_f32_kernel<64, 128, 64>), dim3(tiles), dim3(256), 0, ctx->stream, a);
    RTEN_LAUNCH_CHECK(ctx, "conv_pair_f32_kernel launch");
    return RTEN_HIP_OK;
}

// The same with the first layer's residual computed in the launch: residual = conv1x1(xd, wd) + biasd, a 64-channel pointwise convolution over the same pixels (a stage's
// first block: its shortcut layer).  1 / 0 as above.
RTEN_EXPORT int32_t rten_hip_conv2d_f32_pair_shortcut_supported(const rten_hip_conv2d_desc *d1, const rten_hip_conv2d_desc *ds, const rten_hip_conv2d_desc *d2) {
    if (!ds || !rten_hip_conv2d_f32_pair_supported(d1, d2) || !pointwise_unit(ds)) return 0;
    if (d2->o != 64 || ds->c != 64 || ds->o != d1->o || ds->n != d1->n || ds->h != d1->h || ds->w != d1->w) return 0;
    return 1;
}

RTEN_EXPORT int32_t rten_hip_conv2d_f32_pair_shortcut(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d1, const float *x, const float *w1_packed, const float *bias1,
                                                      const rten_hip_conv2d_desc *ds, const float *xd, const float *wd_packed, const float *biasd, uint32_t flags1, float *y1,
                                                      const rten_hip_conv2d_desc *d2, const float *w2_packed, const float *bias2, uint32_t flags2, float *y2) {
    RTEN_CHECK_CTX(ctx);
    if (!d1 || !ds || !d2 || !x || !w1_packed || !xd || !wd_packed || !w2_packed || !y1 || !y2) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv pair: NULL operand");
    if ((flags1 | flags2) & RTEN_HIP_CONV_RESIDUAL) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair (shortcut form): the residual is the shortcut convolution; no residual tensor");
    if (!rten_hip_conv2d_f32_pair_shortcut_supported(d1, ds, d2)) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair (shortcut form): no one-launch form for these convolutions (rten_hip_conv2d_f32_pair_shortcut_supported)");
    if (((uintptr_t)x | (uintptr_t)xd | (uintptr_t)w1_packed | (uintptr_t)wd_packed | (uintptr_t)w2_packed) & 15) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv pair: operands must be 16-byte aligned");
    const int P = d1->out_h * d1->out_w;
    PairArgs a = {};
    a.X = x; a.W1 = w1_packed; a.B1 = bias1; a.R = nullptr; a.W2 = w2_packed; a.B2 = bias2;
    a.Y1 = y1; a.Y2 = y2;
    a.M1 = d1->o; a.K1 = d1->c; a.M2 = d2->o; a.N = d1->n * P; a.Pn = P;
    a.x_ns = (long long)d1->c * P; a.y1_ns = (long long)d1->o * P; a.y2_ns = (long long)d2->o * P;
    a.w1_cs = (d1->o + 3) & ~3; a.w2_cs = (d2->o + 3) & ~3;
    a.x_bytes = (unsigned)((long long)d1->n * d1->c * P * 4);
    a.w1_bytes = (unsigned)((long long)d1->c * a.w1_cs * 4);
    a.w2_bytes = (unsigned)((long long)d2->c * a.w2_cs * 4);
    a.act1 = (flags1 & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;
    a.act2 = (flags2 & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;
    a.dbg = ctx->debug >> 8;
    a.Xd = xd; a.Wd = wd_packed; a.Bd = biasd;
    a.xd_ns = (long long)ds->c * P; a.wd_cs = (ds->o + 3) & ~3;
    a.xd_bytes = (unsigned)((long long)ds->n * ds->c * P * 4);
    a.wd_bytes = (unsigned)((long long)ds->c * a.wd_cs * 4);
    const int tiles = (a.N + 63) / 64;
    hipLaunchKernelGGL((conv_pair_f32_kernel<64, 64, 64, true>), dim3(tiles), dim3(256), 0, ctx->stream, a);
    RTEN_LAUNCH_CHECK(ctx, "conv_pair_f32_kernel (shortcut form) launch");
    return RTEN_HIP_OK;
}
