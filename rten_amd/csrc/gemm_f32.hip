// MFMA f32 tiled GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// Replaces: GemmExecutor::gemm / gemm_uninit / batched_gemm_uninit (rten-gemm/src/lib.rs:255-372,
// gemm_impl :794-1093), the f32 micro-kernel (kernels/simd_generic.rs:285-414), the virtual
// im2col packing (rten-gemm/src/im2col.rs:56-212, src/ops/conv/im2col.rs:11-128) and conv_impl /
// conv_2d_pointwise (src/ops/conv.rs:33-87,124-365).
//
// One kernel template covers GEMM and convolution:
//     C[m, n] = epilogue( sum_k A[m, k] * B[k, n] )
//   * conv:  m = output channel, k = (c, ky, kx), n = (image, oy, ox) flattened over the WHOLE batch, so
//            late ResNet stages (7x7 / 14x14 maps) still produce thousands of columns per launch.
//            B is never materialised: the im2col gather is fused into the global->LDS tile load.
//            1x1/stride-1 convs skip the gather and read the NCHW tensor as a "two-level" matrix
//            (column n -> image n / P, pixel n % P).
//   * C is written straight into NCHW (row m, two-level column n), with bias / residual Add / Relu /
//     Gelu fused into the epilogue.
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD == the f32 peak, 157 TF).  256 threads = 4
//     waves per workgroup, each wave owns TM x TN accumulator tiles of 32x32.
//   * A and B tiles are staged through LDS k-major ([BK][BM+pad], [BK][BN+pad]) so the MFMA operand
//     fetch (lane -> row k0 + lane/32, column lane%32) is a conflict-free ds_read_b32; tiles are double
//     buffered, the next tile's global loads are issued before the current tile's MFMAs (one barrier
//     per k-tile).
//   * The f32 matrix pipe is slow (64 cycles per MFMA), so the kernel is bound by how few OTHER
//     instructions each wave issues per MFMA.  All global loads are raw buffer loads
//     (buffer_load_dword/dwordx4 ... offen) whose per-lane byte offsets are loop invariant; the k-tile
//     advance rides in the scalar soffset operand, and out-of-tile / out-of-image / k-tail lanes point
//     at an out-of-range offset so the hardware returns 0 -- no exec-mask branches, no 64-bit address
//     arithmetic and no selects in the K loop.  The im2col (c, ky, kx) decomposition comes from a small
//     per-geometry lookup table read with scalar loads (the k row of a wave is uniform).
//   * Workgroup ids are remapped so that each XCD (private L2) owns a contiguous range of tiles that
//     share the same B panel.
//
// Numerics: accumulation order is the reference's, exactly: k-ordered FMA chain per depth block of
// kc = 256 starting from 0, blocks combined with separate adds, bias added after the first block
// (rten-gemm/src/lib.rs:630-633,1008-1013,1221-1255; simd_generic.rs:378-414).  MFMA f32 is a
// k-ordered fmaf chain bit for bit, so outputs are bit-identical to the oracle for M > 1.
#include "gemm_f32_common.h"

namespace {

// MODE: 0 = one depth block, 1 = several depth blocks folded in registers, 2 = split-K producer (see the LDS-DMA kernel).
template <int BM, int BN, int AL, int BL, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_f32_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    constexpr bool MULTI_KC = MODE == 1, SPLIT = MODE == 2;
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_ELEMS = BK * BM / NTHREADS, B_ELEMS = BK * BN / NTHREADS; // per-thread elements per tile
    constexpr int NA = (AL == A_SCALAR) ? A_ELEMS : A_ELEMS / 4;              // loads per thread per tile
    constexpr int NB = (BL == B_SCALAR || BL == B_IM2COL) ? B_ELEMS : B_ELEMS / 4;
    static_assert((NTHREADS / BN) * B_ELEMS == BK, "im2col row mapping must cover the k-tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    float *const As0 = smem;
    float *const Bs0 = smem + 2 * BK * LDA;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;

    // ---- XCD-aware tile mapping: consecutive ids on one XCD walk down a column of tiles (same B panel)
    int tile, grp = -1; // grp >= 0: this workgroup computes one K group of a split tile
    {
        const int nt = gridDim.x;
        const int id = blockIdx.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        if constexpr (SPLIT) {
            const int rr = tile;
            if (p.order & 2) { // K group slowest: an XCD's contiguous id range is one K slice of many tiles
                grp = rr / p.split_ntail;
                tile = p.split_t1 + rr - grp * p.split_ntail;
            } else {           // K group fastest: an XCD's range is all K slices of a few tiles
                tile = p.split_t1 + rr / p.split_s;
                grp = rr - (rr / p.split_s) * p.split_s;
            }
        }
    }
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    // wave-uniform buffer descriptors (kernarg / blockIdx derived only -> SGPRs, no waterfall loops)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);

    // ---- per-thread, loop-invariant byte offsets.  OOB marks lanes outside the tile's valid rows/columns.
    unsigned a_voff[NA];
    int a_krow[NA]; // local k of the element (k-tail test)
    unsigned a_kstep; // byte advance per k-tile (soffset)
    if constexpr (AL == A_M4) { // float4 along m, rows of the K x M (prepacked / transposed) operand
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int idx = t + j * NTHREADS;
            const int k = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
            a_krow[j] = k;
            a_voff[j] = m < p.M ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
        }
        a_kstep = (unsigned)(BK * p.a_cs * 4);
    } else if constexpr (AL == A_K4) { // float4 along k of a row-major [M][K] operand
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int idx = t + j * NTHREADS;
            const int k = (idx % (BK / 4)) * 4, m = m0 + idx / (BK / 4);
            a_krow[j] = k;
            a_voff[j] = m < p.M ? (unsigned)(((long long)m * p.a_rs + k) * 4) : OOB;
        }
        a_kstep = BK * 4;
    } else {
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int idx = t + j * NTHREADS;
            const int k = p.a_dir_m ? idx / BM : idx % BK;
            const int m = m0 + (p.a_dir_m ? idx % BM : idx / BK);
            a_krow[j] = k;
            a_voff[j] = m < p.M ? (unsigned)(((long long)m * p.a_rs + (long long)k * p.a_cs) * 4) : OOB;
        }
        a_kstep = (unsigned)(BK * p.a_cs * 4);
    }

    [[maybe_unused]] unsigned b_voff[NB];
    [[maybe_unused]] int b_krow[NB];
    [[maybe_unused]] unsigned b_kstep = 0;
    [[maybe_unused]] int im_iy0 = 0, im_ix0 = 0, im_pix = 0;
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int idx = t + j * NTHREADS;
            const int k = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(BK * p.b_rs * 4);
    } else if constexpr (BL == B_K4) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int idx = t + j * NTHREADS;
            const int k = (idx % (BK / 4)) * 4, n = n0 + idx / (BK / 4);
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)nb * p.b_ns + (long long)np * p.b_cs + k) * 4) : OOB;
        }
        b_kstep = BK * 4;
    } else if constexpr (BL == B_SCALAR) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int idx = t + j * NTHREADS;
            const int k = p.b_dir_n ? idx / BN : idx % BK;
            const int n = n0 + (p.b_dir_n ? idx % BN : idx / BK);
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + (long long)np * p.b_cs) * 4) : OOB;
        }
        b_kstep = (unsigned)(BK * p.b_rs * 4);
    } else { // B_IM2COL: thread owns column t % BN and rows (t / BN) * B_ELEMS + j (consecutive -> contiguous LUT reads)
        const int n = n0 + (t % BN);
        const bool ok = n < p.N;
        const int nn = ok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const int oy = np / p.OW, ox = np - oy * p.OW;
        im_iy0 = oy * p.sy - p.pt;
        im_ix0 = ox * p.sx - p.pl;
        im_pix = (int)((long long)nb * p.b_ns) + im_iy0 * p.W + im_ix0; // element offset of the (ky=0,kx=0) tap; may be < 0
        if (!ok) im_iy0 = -0x40000000;                                  // fails every bounds test
    }

    float ra[A_ELEMS], rb[B_ELEMS];
    const int nk = (p.K + BK - 1) / BK;

    // im2col LUT entries of the tile that will be prefetched next; read with scalar loads (constant address
    // space + wave-uniform row) one iteration before they are needed, so the gather never waits on them
    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    [[maybe_unused]] i32x2 lutE[B_ELEMS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (BL == B_IM2COL) {
            int krow0 = kt * BK + (t / BN) * B_ELEMS;
            if constexpr (BN >= 64) krow0 = __builtin_amdgcn_readfirstlane(krow0);
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < B_ELEMS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    // ---- global -> register prefetch of k-tile kt (no branches; invalid lanes read 0 through the OOB offset)
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        // the scalar offset never leaves the buffer (the range check subtracts it from num_records): the
        // past-the-end prefetch reuses the last tile's soffset with every lane's voffset out of range
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0);
        const unsigned a_soff = (unsigned)kts * a_kstep;
        const int kleft = p.K - k0; // rows >= kleft are the k tail
        if constexpr (AL == A_SCALAR) {
#pragma unroll
            for (int j = 0; j < NA; j++) ra[j] = buf_load1(rsA, a_krow[j] < kleft ? a_voff[j] : OOB, a_soff);
        } else {
#pragma unroll
            for (int j = 0; j < NA; j++) {
                const f32x4 v = buf_load4(rsA, a_krow[j] < kleft ? a_voff[j] : OOB, a_soff);
                ra[4 * j + 0] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
            }
        }
        if constexpr (BL == B_IM2COL) {
            // virtual im2col row k -> (c, ky, kx) from the LUT entries fetched one iteration ahead
            // (rten-gemm/src/im2col.rs:145-208: out-of-image -> 0)
#pragma unroll
            for (int j = 0; j < B_ELEMS; j++) {
                const i32x2 e = lutE[j];
                const int iy = im_iy0 + (e[1] & 0xffff);
                const int ix = im_ix0 + (e[1] >> 16);
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                rb[j] = buf_load1(rsB, ok ? (unsigned)(im_pix + e[0]) << 2 : OOB, 0);
            }
        } else if constexpr (BL == B_SCALAR) {
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NB; j++) rb[j] = buf_load1(rsB, b_krow[j] < kleft ? b_voff[j] : OOB, b_soff);
        } else {
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const f32x4 v = buf_load4(rsB, b_krow[j] < kleft ? b_voff[j] : OOB, b_soff);
                rb[4 * j + 0] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
            }
        }
    };

    auto store_tile = [&](int buf) {
        float *As = As0 + buf * BK * LDA;
        float *Bs = Bs0 + buf * BK * LDB;
        if constexpr (AL == A_M4) {
#pragma unroll
            for (int j = 0; j < NA; j++) {
                const int idx = t + j * NTHREADS;
                const f32x4 v = {ra[4 * j], ra[4 * j + 1], ra[4 * j + 2], ra[4 * j + 3]};
                *reinterpret_cast<f32x4 *>(As + (idx / (BM / 4)) * LDA + (idx % (BM / 4)) * 4) = v;
            }
        } else if constexpr (AL == A_K4) {
#pragma unroll
            for (int j = 0; j < NA; j++) {
                const int idx = t + j * NTHREADS;
                const int k = (idx % (BK / 4)) * 4, m = idx / (BK / 4);
#pragma unroll
                for (int i = 0; i < 4; i++) As[(k + i) * LDA + m] = ra[4 * j + i];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NA; j++) {
                const int idx = t + j * NTHREADS;
                As[(p.a_dir_m ? idx / BM : idx % BK) * LDA + (p.a_dir_m ? idx % BM : idx / BK)] = ra[j];
            }
        }
        if constexpr (BL == B_N4) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int idx = t + j * NTHREADS;
                const f32x4 v = {rb[4 * j], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]};
                *reinterpret_cast<f32x4 *>(Bs + (idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4) = v;
            }
        } else if constexpr (BL == B_K4) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int idx = t + j * NTHREADS;
                const int k = (idx % (BK / 4)) * 4, n = idx / (BK / 4);
#pragma unroll
                for (int i = 0; i < 4; i++) Bs[(k + i) * LDB + n] = rb[4 * j + i];
            }
        } else if constexpr (BL == B_SCALAR) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int idx = t + j * NTHREADS;
                Bs[(p.b_dir_n ? idx / BN : idx % BK) * LDB + (p.b_dir_n ? idx % BN : idx / BK)] = rb[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < B_ELEMS; j++) Bs[((t / BN) * B_ELEMS + j) * LDB + (t % BN)] = rb[j];
        }
    };

    // ---- accumulators
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    f32x16 acc[TM][TN];
    [[maybe_unused]] f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // flush of one finished depth block into `tot` (between depth blocks, MULTI_KC only)
    // The row/column bases are laundered through an empty asm so that the (rare) flush's address
    // arithmetic is recomputed here instead of being hoisted out of the K loop (~100 live VGPRs).
    [[maybe_unused]] auto flush = [&](bool first) {
        int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
        asm volatile("" : "+v"(mb), "+v"(nb0));
        if (first) fold_first<TM, TN>(p, z, acc, tot, mb, nb0, c_zoff);
        else fold_next<TM, TN>(p, acc, tot);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };

    auto compute_tile = [&](int cur) {
        const float *As = As0 + cur * BK * LDA + wm0 + l31;
        const float *Bs = Bs0 + cur * BK * LDB + wn0 + l31;
        // all MFMA operands of the tile first (ds_read latency overlaps), then the MFMAs back to back
        float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
#pragma unroll
            for (int i = 0; i < TM; i++) af[kk][i] = As[(2 * kk + half) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; j++) bf[kk][j] = Bs[(2 * kk + half) * LDB + j * 32];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_iglp_opt(0);
    };

    [[maybe_unused]] auto store_raw = [&](f32x16 (&v)[TM][TN], int slot) {
        int loff = wave * (TM * TN * 16 * 64) + lane * 4;
        asm volatile("" : "+v"(loff));
        float *base = p.slab + (((long long)z * p.split_ntail + (tile - p.split_t1)) * p.split_slots + slot) * (long long)(BM * BN) + loff;
        if (p.split_counters) { // folded in this launch, possibly on another XCD: write through (see coherent_store4)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base - loff), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 o = {v[i][j][4 * q], v[i][j][4 * q + 1], v[i][j][4 * q + 2], v[i][j][4 * q + 3]};
                        coherent_store4(rs, (unsigned)(loff + ((i * TN + j) * 4 + q) * 256) * 4u, o);
                    }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 o = {v[i][j][4 * q], v[i][j][4 * q + 1], v[i][j][4 * q + 2], v[i][j][4 * q + 3]};
                    *(f32x4 *)(base + ((i * TN + j) * 4 + q) * 256) = o;
                }
    };

    // ---- main loop: depth blocks of KC_TILES k-tiles; inside a block the loop body is branch free
    const int nblk = (MULTI_KC || SPLIT) ? (nk + KC_TILES - 1) / KC_TILES : 1;
    int blk0 = 0, blk1 = nblk;
    if constexpr (SPLIT) {
        blk0 = grp * p.split_g;
        blk1 = blk0 + p.split_g < nblk ? blk0 + p.split_g : nblk;
    }
    const int kt0 = blk0 * KC_TILES; // even: the double-buffer parity of tile kt stays kt & 1
    fetch_lut(kt0);
    load_tile(kt0);
    fetch_lut(kt0 + 1);
    store_tile(0);
    __syncthreads();
    for (int blk = blk0; blk < blk1; blk++) {
        const int kt_end = (MULTI_KC || SPLIT) ? ((blk + 1) * KC_TILES < nk ? (blk + 1) * KC_TILES : nk) : nk;
        for (int kt = blk * KC_TILES; kt < kt_end; kt++) {
            load_tile(kt + 1); // prefetch; past the end every lane is out of range -> zeros, never used
            fetch_lut(kt + 2);
            compute_tile(kt & 1);
            store_tile((kt + 1) & 1);
            __syncthreads();
        }
        if constexpr (SPLIT) {
            store_raw(acc, blk);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        }
        if constexpr (MULTI_KC) {
            if (blk + 1 < nblk) flush(blk == 0);
        }
    }

    // ---- final depth block + fused epilogue (residual Add, activation), NCHW / row-major store
    if constexpr (!SPLIT) {
        if (!(ABLATE(p) & 4)) {
            const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
            if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
                fold_next<TM, TN>(p, acc, tot);
                store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
            } else {
                fold_first<TM, TN>(p, z, acc, acc, mb, nb0, c_zoff);
                store_out<TM, TN>(p, acc, mb, nb0, c_zoff);
            }
        }
    } else if (p.split_counters) {
        split_finish<BM, BN, TM, TN>(p, z, tile, wave, lane, m0, n0, c_zoff, reinterpret_cast<int *>(smem));
    }
}


// =====================================================================================================
// LDS-DMA variant (conv paths: A = prepacked [K][M] weights, B = dense two-level or im2col gather).
//
// Tiles go HBM/L2 -> LDS directly (`buffer_load_dword[x4] ... offen lds`): no staging VGPRs, no ds_write
// pass, and three LDS stages keep two k-tiles in flight behind the one being multiplied, so the
// ~1200-cycle load latency hides under the matrix pipe even when only one or two workgroups fit on a CU.
// Per k-tile: counted `s_waitcnt vmcnt(N)` (never 0 inside the loop) -> raw s_barrier -> issue the DMA of
// tile kt+2 into the stage that was just freed -> MFMAs of tile kt (operand fragments double buffered in
// registers so ds_read latency overlaps the previous MFMA group).  All LDS lives in ONE __shared__
// array (a second object would make hipcc drain vmcnt before every ds_read -- cdna_hip_programming.md).
// LDS image: As[BK][BM], Bs[BK][BN] unpadded (DMA writes are lane-linear); MFMA operand reads walk
// consecutive columns, so they are conflict-free without padding.
// =====================================================================================================
// LDS stages per tile shape (measured: 4-6 stages cost occupancy and do not speed up a lone workgroup)
constexpr int nstage_for(int bm, int bn) { return 3; } // deeper rings measured slower: LDS-limited occupancy, no gain for a lone workgroup
constexpr int MAX_NSTAGE = 6;

// MODE 0: K <= 256 (one depth block); 1: several depth blocks folded in registers; 2: split-K producer -- every
// workgroup computes one group of depth blocks of one split tile and parks each block's raw accumulator in the slab
// (no fold, no epilogue: igemm_f32_fixup_kernel finishes the tile).
// AL: A_M4 = k-major A ([K][M], prepacked conv weights, transposed GEMM operands); A_K4 = row-major A ([M][K], the
// plain MatMul layout): one DMA instruction then moves 64 rows x one k-quad and the LDS image is [k-quad][m][4].
// MFK: 0 = operand fragments double buffered across k-pairs with the next pair's ds_reads behind the current MFMA group (iglp_opt);
//      1 = all of the k-tile's fragments first, then the MFMAs back to back with nothing between them: with one 32x32 block per
//          wave (64x64 tiles) consecutive MFMAs hit the SAME accumulator, and any instruction issued between two such MFMAs
//          costs ~43 cycles on top of its own slot (MI355X_MICROARCH.md, per-instruction constants).
// Waves per SIMD the compiler must leave room for: 64x64 tiles are LDS-limited to 6 (three stages) / 10 (two stages) workgroups per compute unit,
// so their register budget is set to match (80 VGPRs: the MODE 1 / 2 forms sat at 81-85, i.e. at 5) -- more resident workgroups is what these
// kernels respond to (tools/debug/f32_trace.py with RTEN_HIP_OCC_CAP: 2 -> 3 -> 6 workgroups per CU = 3.72 -> 3.20 -> 2.95 ms per step).
// (four stages of a 64x64 tile are 32 KB: LDS admits four to five workgroups per compute unit, so asking the compiler for a six-wave register budget only made it
// report an unmet target -- rounds 3-5; the four-stage forms now ask for what they can have)
constexpr int dma_min_waves(int bm, int bn, int mode, int nst = 3) { return bm * bn == 64 * 64 ? ((mode == 3 || nst >= 4) ? 4 : 6) : 2; }
template <int BM, int BN, int AL, int BL, int MODE, int NST = 3, int MFK = 0>
__global__ __launch_bounds__(NTHREADS, dma_min_waves(BM, BN, MODE, NST)) void igemm_f32_dma_kernel(const GemmArgs p) {
    TR_DECL
    TR_STAMP(0)
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    // MODE 3 ("mixed"): one launch holds the whole tiles [0, split_t1) (fold + epilogue, as MODE 1) AND the split-K
    // producers of the tail tiles (as MODE 2), so the tail's small workgroups fill the last round next to the whole
    // tiles instead of running alone afterwards.
    constexpr bool MIXED = MODE == 3, MULTI_KC = MODE == 1 || MIXED, SPLIT = MODE == 2;
    static_assert(AL == A_M4 || AL == A_K4, "DMA kernel: A is k-major or row-major with 16-byte rows");
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "DMA kernel covers the conv operand layouts");
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STAGE = BK * (BM + BN); // floats per stage
    constexpr int NA = BK * BM / 256 / 4; // dwordx4 DMA instructions per wave per tile (A)
    constexpr int NBV = BK * BN / 256 / 4; // dwordx4 (dense B)
    constexpr int NBG = BK * BN / 64 / 4;  // dword gathers per wave per tile (im2col B)
    constexpr int PER_TILE = NA + (BL == B_N4 ? NBV : NBG);
    static_assert(NA >= 1 && NBV >= 1, "tile too small for 4-wave DMA split");
    constexpr int NSTAGE = NST;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;

    int tile, grp = -1; // grp >= 0: this workgroup computes one K group of a split tile
    {
        const int id = blockIdx.x;
        const int nt = MIXED ? p.split_t1 : (int)gridDim.x; // whole tiles are XCD-chunked; mixed-mode producers keep dispatch order
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        if (SPLIT || (MIXED && id >= p.split_t1)) {
            const int rr = MIXED ? id - p.split_t1 : tile;
            if (p.order & 2) { // K group slowest: an XCD's contiguous id range is one K slice of many tiles
                grp = rr / p.split_ntail;
                tile = p.split_t1 + rr - grp * p.split_ntail;
            } else {           // K group fastest: an XCD's range is all K slices of a few tiles
                tile = p.split_t1 + rr / p.split_s;
                grp = rr - (rr / p.split_s) * p.split_s;
            }
        }
    }
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + BK - 1) / BK;

    // ---- loop-invariant DMA source offsets.  Wave w issues instructions q = w*N + j; instruction q covers
    // the flat tile range [q*256, q*256+256) floats (dwordx4) or [q*64, q*64+64) (dword gather).
    unsigned a_voff[NA];
    [[maybe_unused]] int a_kq[NA]; // A_K4: first local k of the instruction's k-quad (k-tail test)
#pragma unroll
    for (int j = 0; j < NA; j++) {
        if constexpr (AL == A_M4) {
            const int f = (wave * NA + j) * 256 + lane * 4;
            const int k = f / BM, m = m0 + f % BM;
            // rows >= K lie past the end of the [K][M4] buffer (hardware range check); columns >= M4 must not wrap
            a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
        } else {
            const int q = wave * NA + j, kq = q / (BM / 64), m = m0 + (q % (BM / 64)) * 64 + lane;
            a_kq[j] = kq * 4;
            a_voff[j] = m < p.M ? (unsigned)(((long long)m * p.a_rs + kq * 4) * 4) : OOB;
        }
    }
    const unsigned a_kstep = AL == A_M4 ? (unsigned)(BK * p.a_cs * 4) : (unsigned)(BK * 4);

    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] unsigned b_kstep = 0;
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    constexpr int NCOL = IM2COL && BN == 128 ? 2 : 1;
    [[maybe_unused]] int im_iy0[NCOL], im_ix0[NCOL], im_pix[NCOL];
    [[maybe_unused]] unsigned im_inv[NCOL]; // TAPS: bit t set = tap t of this lane's pixel is padding; bit 31 always set (k-tail rows)
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NBV; j++) {
            const int f = (wave * NBV + j) * 256 + lane * 4;
            const int k = f / BN, n = n0 + f % BN;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(BK * p.b_rs * 4);
    } else {
        // gather instruction q = wave*NBG + j covers row q / (BN/64), columns (q % (BN/64))*64 + lane.
        // A lane therefore sees at most BN/64 distinct columns.
#pragma unroll
        for (int c = 0; c < BN / 64; c++) {
            const int n = n0 + c * 64 + lane;
            const bool ok = n < p.N;
            const int nn = ok ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            const int oy = np / p.OW, ox = np - oy * p.OW;
            im_iy0[c] = ok ? oy * p.sy - p.pt : -0x40000000;
            im_ix0[c] = ox * p.sx - p.pl;
            im_pix[c] = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0[c];
            if constexpr (TAPS) {
                unsigned colbad = 0; // bit kx set: column tap kx falls outside the image
                for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0[c] + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
                const unsigned allbad = (1u << p.KW) - 1u;
                unsigned inv = 0x80000000u;
                for (int ky = 0; ky < p.KH; ky++)
                    inv |= ((unsigned)(im_iy0[c] + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
                im_inv[c] = inv;
            }
        }
    }

    // im2col LUT entries (scalar loads) for the tile whose DMA is issued NEXT
    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    constexpr int LROWS = BK / 4; // rows of a tile handled by one wave (NBG / (BN/64))
    [[maybe_unused]] i32x2 lutE[LROWS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (IM2COL) {
            const int krow0 = kt * BK + wave * LROWS;
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < LROWS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + BK * BM;
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0); // keep the scalar offset inside the buffer
        const bool past = kt >= nk;
        const unsigned a_soff = (unsigned)kts * a_kstep;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            bool dead = past;
            if constexpr (AL == A_K4) dead = a_kq[j] >= p.K - kt * BK; // k-tail quads (and every quad past the end) read as zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + (wave * NA + j) * 256), 16,
                                                     (int)(dead ? OOB : a_voff[j]), (int)a_soff, 0, 0);
        }
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * BK;
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBV + j) * 256), 16,
                                                         (int)(b_krow[j] < kleft ? b_voff[j] : OOB), (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NBG; j++) {
                constexpr int CPR = BN / 64;            // gather instructions per tile row
                const int r = j / CPR, c = j % CPR;     // row within this wave's LROWS, column chunk
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    // e[1] = 31 - tap: the tap's padding bit moves to bit 31 and pushes the offset out of range
                    voff = ((im_inv[c] << e[1]) & 0x80000000u) | ((unsigned)(im_pix[c] + e[0]) << 2);
                } else {
                    const int iy = im_iy0[c] + (e[1] & 0xffff);
                    const int ix = im_ix0[c] + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix[c] + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * LROWS + r) * BN + c * 64), 4, (int)voff, 0, 0, 0);
            }
        }
    };

    // ---- accumulators / epilogue helpers (same numerics as igemm_f32_kernel)
    const int wq = t >> 6; // per-lane copy of the wave id for address math
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    f32x16 acc[TM][TN];
    [[maybe_unused]] f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    [[maybe_unused]] auto flush = [&](bool first) {
        int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
        asm volatile("" : "+v"(mb), "+v"(nb0));
        if (first) fold_first<TM, TN>(p, z, acc, tot, mb, nb0, c_zoff);
        else fold_next<TM, TN>(p, acc, tot);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };

    auto compute_tile = [&](int stage) {
        // A fragment of k-pair kk, block i: k = 2*kk + half.  k-major image: As[k][m]; row-major image: [k/4][m][4]
        // (k and k+1 share a quad, so `half` is part of the lane's base and the rest is an immediate).
        const float *As = smem + stage * STAGE + (AL == A_M4 ? wm0 + l31 + half * BM : (wm0 + l31) * 4 + half);
        auto a_idx = [](int kk, int i) { return AL == A_M4 ? 2 * kk * BM + i * 32 : (kk >> 1) * BM * 4 + ((2 * kk) & 3) + i * 128; };
        const float *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
        if constexpr (MFK == 1) {
            float afa[BK / 2][TM], bfa[BK / 2][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) {
#pragma unroll
                for (int i = 0; i < TM; i++) afa[kk][i] = As[a_idx(kk, i)];
#pragma unroll
                for (int j = 0; j < TN; j++) bfa[kk][j] = Bs[(2 * kk + half) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0); // every ds_read of the tile is issued before the first MFMA
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afa[kk][i], bfa[kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        float af[2][TM], bf[2][TN]; // operand fragments, double buffered across k-pairs
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = As[a_idx(0, i)];
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = Bs[half * BN + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[nxt][i] = As[a_idx(kk + 1, i)];
#pragma unroll
                for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(2 * (kk + 1) + half) * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
    };

    // raw accumulator image of this workgroup's tile in the split-K slab: [wave][i][j][quad][lane] float4
    [[maybe_unused]] auto store_raw = [&](f32x16 (&v)[TM][TN], int slot) {
        int loff = wq * (TM * TN * 16 * 64) + lane * 4;
        asm volatile("" : "+v"(loff)); // keep the address math at the use (not hoisted across the K loop)
        float *base = p.slab + (((long long)z * p.split_ntail + (tile - p.split_t1)) * p.split_slots + slot) * (long long)(BM * BN) + loff;
        if (p.split_counters) { // folded in this launch, possibly on another XCD: write through (see coherent_store4)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base - loff), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 o = {v[i][j][4 * q], v[i][j][4 * q + 1], v[i][j][4 * q + 2], v[i][j][4 * q + 3]};
                        coherent_store4(rs, (unsigned)(loff + ((i * TN + j) * 4 + q) * 256) * 4u, o);
                    }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 o = {v[i][j][4 * q], v[i][j][4 * q + 1], v[i][j][4 * q + 2], v[i][j][4 * q + 3]};
                    *(f32x4 *)(base + ((i * TN + j) * 4 + q) * 256) = o;
                }
    };

    // ---- software pipeline: tiles kt+1 and kt+2 are in flight while tile kt is multiplied
    const int nblk = (MULTI_KC || SPLIT) ? (nk + KC_TILES - 1) / KC_TILES : 1;
    int blk0 = 0, blk1 = nblk;
    if (SPLIT || (MIXED && grp >= 0)) {
        blk0 = grp * p.split_g;
        blk1 = blk0 + p.split_g < nblk ? blk0 + p.split_g : nblk;
    }
    const int kt0 = blk0 * KC_TILES;
    fetch_lut(kt0);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) {
        issue_tile(kt0 + i, i);
        fetch_lut(kt0 + i + 1);
    }
    int stage = 0;
    TR_STAMP(1)
    for (int blk = blk0; blk < blk1; blk++) {
        const int kt_end = (MULTI_KC || SPLIT) ? ((blk + 1) * KC_TILES < nk ? (blk + 1) * KC_TILES : nk) : nk;
        for (int kt = blk * KC_TILES; kt < kt_end; kt++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>(); // this wave's DMA for tile kt has landed; NSTAGE-2 later tiles stay in flight
            if (!(ABLATE(p) & 8)) __builtin_amdgcn_s_barrier(); // ... and everyone else's; all waves are done reading the stage of tile kt-1
#ifdef RTEN_TRACE
            if (tr_trips == 0) TR_STAMP(2)
            tr_trips++;
#endif
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1; // (kt + NSTAGE - 1) % NSTAGE: the stage tile kt-1 used
            if (!(ABLATE(p) & 1)) issue_tile(kt + NSTAGE - 1, stp);
            fetch_lut(kt + NSTAGE);
            if (ABLATE(p) & 16) { // ablation: MFMAs on register operands only (no ds_read)
                float fa = (float)kt, fb = (float)lane;
#pragma unroll
                for (int kk = 0; kk < BK / 2; kk++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i][j], 0, 0, 0);
            } else if (!(ABLATE(p) & 2)) compute_tile(stage); // (s_setprio around the matrix phase: measured, no gain)
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        if (SPLIT || (MIXED && grp >= 0)) {
            // order bit 3 (RELAXED split-K, measurement only, NOT the reference's order): the group's depth blocks accumulate in one register block and ONE
            // partial per group is parked (slot = group) -- what an order-free split-K would move; the strict form parks every depth block
            const bool relaxed = (p.order & 8) != 0;
            if (!relaxed || blk + 1 == blk1) {
                store_raw(acc, relaxed ? grp : blk);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            }
        } else if constexpr (MULTI_KC) {
            if (blk + 1 < nblk) flush(blk == 0);
        }
    }
    wait_vmcnt<0>(); // drain the two look-ahead tiles before the LDS goes away
    TR_STAMP(3)
    [[maybe_unused]] constexpr unsigned TR_KID = BM | (BN << 8) | (MODE << 16) | (BL << 20) | (AL << 24) | (NST << 28);

    if (SPLIT || (MIXED && grp >= 0)) {
        if (p.split_counters) split_finish<BM, BN, TM, TN>(p, z, tile, wq, lane, m0, n0, c_zoff, reinterpret_cast<int *>(smem));
        TR_STAMP(4)
        TR_WRITE(TR_KID, tile, grp)
        return;
    }
    if constexpr (!SPLIT) {
        if (!(ABLATE(p) & 4)) {
            const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
            if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
                fold_next<TM, TN>(p, acc, tot);
                store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
            } else {
                fold_first<TM, TN>(p, z, acc, acc, mb, nb0, c_zoff);
                store_out<TM, TN>(p, acc, mb, nb0, c_zoff);
            }
        }
    }
    TR_STAMP(4)
    TR_WRITE(TR_KID, tile, grp)
}

// Split-K fixup: one WAVE per quadrant of a split tile (grid = 4 x split tiles, 64 threads), same lane <-> element
// mapping as the GEMM kernels.  Replays the unsplit kernel's fold over the parked per-block accumulators in
// depth-block order (first block: beta*C + bias; later blocks: separate adds), then the shared epilogue (residual,
// activation, store).  Slots are fetched U at a time so several loads are in flight per lane.
template <int BM, int BN>
__global__ __launch_bounds__(64) void igemm_f32_fixup_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int U = TM * TN >= 4 ? 1 : 4 / (TM * TN); // slots per batch: 64 floats per lane in flight
    const int lane = threadIdx.x, wq = blockIdx.x & 3, ti = blockIdx.x >> 2;
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;
    const int tile = p.split_t1 + ti;
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;
    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    const float *base = p.slab + ((long long)z * p.split_ntail + ti) * p.split_slots * (long long)(BM * BN) +
                        wq * (TM * TN * 16 * 64) + lane * 4;
    f32x16 acc[U][TM][TN], tot[TM][TN];
    auto load_raw = [&](f32x16 (&v)[TM][TN], int slot) {
        const float *b = base + (long long)slot * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 o = *(const f32x4 *)(b + ((i * TN + j) * 4 + q) * 256);
                    v[i][j][4 * q] = o[0]; v[i][j][4 * q + 1] = o[1]; v[i][j][4 * q + 2] = o[2]; v[i][j][4 * q + 3] = o[3];
                }
    };
    const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
    load_raw(acc[0], 0);
    fold_first<TM, TN>(p, z, acc[0], tot, mb, nb0, c_zoff);
    int s = 1;
    for (; s + U <= p.split_slots; s += U) {
#pragma unroll
        for (int u = 0; u < U; u++) load_raw(acc[u], s + u);
#pragma unroll
        for (int u = 0; u < U; u++) fold_next<TM, TN>(p, acc[u], tot);
    }
    for (; s < p.split_slots; s++) {
        load_raw(acc[0], s);
        fold_next<TM, TN>(p, acc[0], tot);
    }
    store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
}

// =====================================================================================================
// LDS-DMA kernel on v_mfma_f32_16x16x4_f32: the same tile DMA, LDS image and depth-block fold as igemm_f32_dma_kernel, but a
// wave's (BM/2) x (BN/2) share is a grid of 16x16 accumulator blocks.  With 64x64 tiles the 32x32x2 form leaves every wave ONE
// accumulator, i.e. a chain of dependent MFMAs: whatever the wave issues between two of them (the next operands' ds_reads) costs
// ~43 cycles beyond its own slot, and the dependent latency itself is the whole 64-cycle issue time.  Four independent 16x16
// accumulators (40-cycle dependent latency, revisited every 128 cycles) take both stalls away at the same LDS traffic per flop.
// v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (tools/probes/mfma_16x16x4_order.hip): same bits.  MODE 0 / 1 (no split-K form).
// Accumulator element r of block (i, j): row wm0 + 16 i + 4 * (lane / 16) + r, column wn0 + 16 j + lane % 16.
// =====================================================================================================
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int TM2, int TN2>
__device__ __forceinline__ void fold_first16(const GemmArgs &p, int z, f32x4v (&acc)[TM2][TN2], f32x4v (&out)[TM2][TN2], int mb, int nb0, long long c_zoff) {
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void *)((p.bias ? p.bias : p.C) + (long long)z * p.bias_bs), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + c_zoff), 0, 0x7ffffffc, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM2; i++) {
        const int mrow = mb + i * 16;
        float brow[4];
        if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) {
#pragma unroll
            for (int r = 0; r < 4; r++) brow[r] = buf_load1(rsBias, mrow + r < p.M ? (unsigned)(mrow + r) << 2 : OOB, 0);
        }
#pragma unroll
        for (int j = 0; j < TN2; j++) {
            const int n = nb0 + j * 16;
            const bool cok = n < p.N;
            f32x4v v = acc[i][j];
            if (p.beta == 0.f) {
                if (p.alpha != 1.f) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = v[r] * p.alpha;
                }
            } else {
                const int nn = cok ? n : 0;
                const int nb = nn / p.Pn, np = nn - nb * p.Pn;
                const unsigned col = (unsigned)((long long)nb * p.c_ns + np);
                float cin[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = mrow + r;
                    cin[r] = buf_load1(rsC, (m < p.M && cok) ? (col + (unsigned)m * (unsigned)p.c_rs) << 2 : OOB, 0);
                }
                if (p.beta == 1.f && p.alpha == 1.f) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = cin[r] + v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = vm::fma(v[r], p.alpha, cin[r] * p.beta);
                }
            }
            if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = v[r] + brow[r];
            } else if (p.bias_kind == RTEN_HIP_BIAS_PER_COL) {
                const float bcol = buf_load1(rsBias, cok ? (unsigned)n << 2 : OOB, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = v[r] + bcol;
            }
            out[i][j] = v;
        }
    }
}

template <int TM2, int TN2>
__device__ __forceinline__ void fold_next16(const GemmArgs &p, f32x4v (&acc)[TM2][TN2], f32x4v (&tot)[TM2][TN2]) {
#pragma unroll
    for (int i = 0; i < TM2; i++)
#pragma unroll
        for (int j = 0; j < TN2; j++) {
            if (p.alpha == 1.f) {
#pragma unroll
                for (int r = 0; r < 4; r++) tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) tot[i][j][r] = vm::fma(acc[i][j][r], p.alpha, tot[i][j][r]);
            }
        }
}

template <int TM2, int TN2>
__device__ __forceinline__ void store_out16(const GemmArgs &p, f32x4v (&val)[TM2][TN2], int mb, int nb0, long long c_zoff) {
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + c_zoff), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)((p.res ? p.res : p.C) + c_zoff), 0, 0x7ffffffc, 0x00020000);
    const bool has_res = p.res != nullptr;
    const unsigned rs4 = (unsigned)p.c_rs << 2;
#pragma unroll
    for (int j = 0; j < TN2; j++) {
        const int n = nb0 + j * 16;
        const bool cok = n < p.N;
        const int nn = cok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const unsigned col = (unsigned)((long long)nb * p.c_ns + np);
        unsigned voff[TM2][4];
        float rr[TM2][4];
#pragma unroll
        for (int i = 0; i < TM2; i++) {
            const int mrow = mb + i * 16;
            const unsigned base = cok ? (col + (unsigned)mrow * (unsigned)p.c_rs) << 2 : OOB;
#pragma unroll
            for (int r = 0; r < 4; r++) voff[i][r] = mrow + r < p.M ? base : OOB;
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 4; r++) rr[i][r] = buf_load1(rsR, voff[i][r], (unsigned)r * rs4);
            }
        }
#pragma unroll
        for (int i = 0; i < TM2; i++) {
            f32x4v v = val[i][j];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = v[r] + rr[i][r];
            }
            if (p.act == RTEN_HIP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = vm::relu(v[r]);
            } else if (p.act == RTEN_HIP_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = vm::gelu(v[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float x = v[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rsC, (int)voff[i][r], (int)((unsigned)r * rs4), 0);
            }
        }
    }
}

template <int BM, int BN, int AL, int BL, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_f32_dma16_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    constexpr bool MULTI_KC = MODE == 1;
    static_assert(MODE == 0 || MODE == 1, "the 16x16x4 kernel has no split-K form");
    static_assert(AL == A_M4 || AL == A_K4, "DMA kernel: A is k-major or row-major with 16-byte rows");
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "DMA kernel covers the conv operand layouts");
    constexpr int WM = 2, WN = 2;
    constexpr int TM2 = BM / WM / 16, TN2 = BN / WN / 16;
    constexpr int STAGE = BK * (BM + BN);
    constexpr int NA = BK * BM / 256 / 4;
    constexpr int NBV = BK * BN / 256 / 4;
    constexpr int NBG = BK * BN / 64 / 4;
    constexpr int PER_TILE = NA + (BL == B_N4 ? NBV : NBG);
    constexpr int NSTAGE = 3;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    const int z = blockIdx.y;

    int tile;
    {
        const int id = blockIdx.x;
        const int nt = (int)gridDim.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + BK - 1) / BK;

    unsigned a_voff[NA];
    [[maybe_unused]] int a_kq[NA];
#pragma unroll
    for (int j = 0; j < NA; j++) {
        if constexpr (AL == A_M4) {
            const int f = (wave * NA + j) * 256 + lane * 4;
            const int k = f / BM, m = m0 + f % BM;
            a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
        } else {
            const int q = wave * NA + j, kq = q / (BM / 64), m = m0 + (q % (BM / 64)) * 64 + lane;
            a_kq[j] = kq * 4;
            a_voff[j] = m < p.M ? (unsigned)(((long long)m * p.a_rs + kq * 4) * 4) : OOB;
        }
    }
    const unsigned a_kstep = AL == A_M4 ? (unsigned)(BK * p.a_cs * 4) : (unsigned)(BK * 4);

    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] unsigned b_kstep = 0;
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    constexpr int NCOL = IM2COL && BN == 128 ? 2 : 1;
    [[maybe_unused]] int im_iy0[NCOL], im_ix0[NCOL], im_pix[NCOL];
    [[maybe_unused]] unsigned im_inv[NCOL];
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NBV; j++) {
            const int f = (wave * NBV + j) * 256 + lane * 4;
            const int k = f / BN, n = n0 + f % BN;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(BK * p.b_rs * 4);
    } else {
#pragma unroll
        for (int c = 0; c < BN / 64; c++) {
            const int n = n0 + c * 64 + lane;
            const bool ok = n < p.N;
            const int nn = ok ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            const int oy = np / p.OW, ox = np - oy * p.OW;
            im_iy0[c] = ok ? oy * p.sy - p.pt : -0x40000000;
            im_ix0[c] = ox * p.sx - p.pl;
            im_pix[c] = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0[c];
            if constexpr (TAPS) {
                unsigned colbad = 0;
                for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0[c] + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
                const unsigned allbad = (1u << p.KW) - 1u;
                unsigned inv = 0x80000000u;
                for (int ky = 0; ky < p.KH; ky++)
                    inv |= ((unsigned)(im_iy0[c] + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
                im_inv[c] = inv;
            }
        }
    }

    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    constexpr int LROWS = BK / 4;
    [[maybe_unused]] i32x2 lutE[LROWS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (IM2COL) {
            const int krow0 = kt * BK + wave * LROWS;
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < LROWS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + BK * BM;
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0);
        const bool past = kt >= nk;
        const unsigned a_soff = (unsigned)kts * a_kstep;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            bool dead = past;
            if constexpr (AL == A_K4) dead = a_kq[j] >= p.K - kt * BK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + (wave * NA + j) * 256), 16, (int)(dead ? OOB : a_voff[j]), (int)a_soff, 0, 0);
        }
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * BK;
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBV + j) * 256), 16, (int)(b_krow[j] < kleft ? b_voff[j] : OOB), (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NBG; j++) {
                constexpr int CPR = BN / 64;
                const int r = j / CPR, c = j % CPR;
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    voff = ((im_inv[c] << e[1]) & 0x80000000u) | ((unsigned)(im_pix[c] + e[0]) << 2);
                } else {
                    const int iy = im_iy0[c] + (e[1] & 0xffff);
                    const int ix = im_ix0[c] + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix[c] + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * LROWS + r) * BN + c * 64), 4, (int)voff, 0, 0, 0);
            }
        }
    };

    const int wq = t >> 6;
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    f32x4v acc[TM2][TN2];
    [[maybe_unused]] f32x4v tot[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; i++)
#pragma unroll
        for (int j = 0; j < TN2; j++) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] auto flush = [&](bool first) {
        int mb = m0 + wm0 + 4 * quad, nb0 = n0 + wn0 + l15;
        asm volatile("" : "+v"(mb), "+v"(nb0));
        if (first) fold_first16<TM2, TN2>(p, z, acc, tot, mb, nb0, c_zoff);
        else fold_next16<TM2, TN2>(p, acc, tot);
#pragma unroll
        for (int i = 0; i < TM2; i++)
#pragma unroll
            for (int j = 0; j < TN2; j++) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };

    auto compute_tile = [&](int stage) {
        // k-step ks covers rows 4 ks .. 4 ks + 3 of the tile; lane -> k = 4 ks + quad.  k-major image As[k][m]; row-major image [k/4][m][4]
        const float *As = smem + stage * STAGE + (AL == A_M4 ? wm0 + l15 + quad * BM : (wm0 + l15) * 4 + quad);
        auto a_idx = [](int ks, int i) { return AL == A_M4 ? 4 * ks * BM + i * 16 : ks * BM * 4 + i * 64; };
        const float *Bs = smem + stage * STAGE + BK * BM + wn0 + l15 + quad * BN;
        float af[2][TM2], bf[2][TN2];
#pragma unroll
        for (int i = 0; i < TM2; i++) af[0][i] = As[a_idx(0, i)];
#pragma unroll
        for (int j = 0; j < TN2; j++) bf[0][j] = Bs[j * 16];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ks++) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 4) {
#pragma unroll
                for (int i = 0; i < TM2; i++) af[nxt][i] = As[a_idx(ks + 1, i)];
#pragma unroll
                for (int j = 0; j < TN2; j++) bf[nxt][j] = Bs[4 * (ks + 1) * BN + j * 16];
            }
#pragma unroll
            for (int i = 0; i < TM2; i++)
#pragma unroll
                for (int j = 0; j < TN2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
    };

    const int nblk = MULTI_KC ? (nk + KC_TILES - 1) / KC_TILES : 1;
    fetch_lut(0);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) {
        issue_tile(i, i);
        fetch_lut(i + 1);
    }
    int stage = 0;
    for (int blk = 0; blk < nblk; blk++) {
        const int kt_end = MULTI_KC ? ((blk + 1) * KC_TILES < nk ? (blk + 1) * KC_TILES : nk) : nk;
        for (int kt = blk * KC_TILES; kt < kt_end; kt++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>();
            __builtin_amdgcn_s_barrier();
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            issue_tile(kt + NSTAGE - 1, stp);
            fetch_lut(kt + NSTAGE);
            compute_tile(stage);
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        if constexpr (MULTI_KC) {
            if (blk + 1 < nblk) flush(blk == 0);
        }
    }
    wait_vmcnt<0>();

    const int mb = m0 + wm0 + 4 * quad, nb0 = n0 + wn0 + l15;
    if constexpr (MULTI_KC) {
        fold_next16<TM2, TN2>(p, acc, tot);
        store_out16<TM2, TN2>(p, tot, mb, nb0, c_zoff);
    } else {
        fold_first16<TM2, TN2>(p, z, acc, acc, mb, nb0, c_zoff);
        store_out16<TM2, TN2>(p, acc, mb, nb0, c_zoff);
    }
}

// =====================================================================================================
// Persistent LDS-DMA kernel: a workgroup walks a LIST of tiles and its tile DMA runs two k-tiles ahead ACROSS tile boundaries.
//
// Why: a pure MFMA stream sustains 154.7 TFLOP/s on this chip (tools/probes/mfma_sustained.hip: 2381 MHz under load), yet the
// one-tile-per-workgroup kernels above reach 80-105 on ResNet's layers.  Their tiles are short (K = 64 ... 576: 7 us of matrix
// work) and every workgroup starts with ~2 us of load latency and ends with an epilogue that waits on its residual loads and
// stores; all resident workgroups of a CU begin together and stay in lockstep, so those phases do not overlap anybody's MFMAs,
// and the partial last round costs a whole tile latency.  Here a CU's resident workgroups live for the whole launch:
//   * the loader (same DMA instructions, same LDS ring) keeps its own (tile, k-tile) position and simply continues into the next
//     tile of the list, recomputing its per-lane source offsets when it crosses -- the first k-tiles of tile i+1 land while tile
//     i's last MFMAs and epilogue run: no load bubble between tiles;
//   * the launch-time prologue (kernarg loads, LUT warm-up, first DMA latency) is paid once per workgroup, not once per tile;
//   * the grid is num_cus x R workgroups (R = split `groups` of plan mode 5), tiles are dealt round-robin inside each XCD's
//     contiguous chunk (same L2 locality as the remapped ids above).
// Numerics: per output element exactly the chain of the other kernels (k-ordered MFMA chain per depth block of 256, blocks
// folded with separate adds, bias after the first block): bit-identical.  MF16 selects v_mfma_f32_16x16x4_f32 blocks.
// =====================================================================================================
template <int BM, int BN, int AL, int BL, bool MF16, int MFK = 0>
__global__ __launch_bounds__(NTHREADS, (BM * BN >= 128 * 128) ? 1 : 2) void igemm_f32_pers_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    static_assert(AL == A_M4 || AL == A_K4, "DMA kernel: A is k-major or row-major with 16-byte rows");
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "DMA kernel covers the conv operand layouts");
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;    // 32x32 blocks per wave
    constexpr int TM2 = BM / WM / 16, TN2 = BN / WN / 16;  // 16x16 blocks per wave
    constexpr int STAGE = BK * (BM + BN);
    constexpr int NA = BK * BM / 256 / 4;
    constexpr int NBV = BK * BN / 256 / 4;
    constexpr int NBG = BK * BN / 64 / 4;
    constexpr int PER_TILE = NA + (BL == B_N4 ? NBV : NBG);
    constexpr int NSTAGE = 3;
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    constexpr int NCOL = IM2COL && BN == 128 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int z = blockIdx.y;

    // ---- this workgroup's tile list: XCD x = id & 7 owns the contiguous chunk [lo, lo + cnt) of the launch's tiles (as the
    // remapped ids of the one-tile kernels), its workgroups j = id >> 3 take tiles lo + j, lo + j + gx, ...
    const int T = p.tiles_m * p.tiles_n;
    int t_next, t_end, t_step;
    {
        const int id = blockIdx.x, G = (int)gridDim.x;
        const int xcd = id & 7, q = T >> 3, r = T & 7;
        const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        t_step = (G - xcd + 7) >> 3; // workgroups on this XCD
        t_next = lo + (id >> 3);
        t_end = lo + cnt;
    }
    if (t_next >= t_end) return; // more workgroups than tiles on this XCD (uniform per workgroup: no barrier is skipped by part of it)

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + BK - 1) / BK;
    const unsigned a_kstep = AL == A_M4 ? (unsigned)(BK * p.a_cs * 4) : (unsigned)(BK * 4);
    const unsigned b_kstep = BL == B_N4 ? (unsigned)(BK * p.b_rs * 4) : 0u;

    // ---- loader state: position (l_tile, l_kt) in this workgroup's stream of k-tiles and the per-lane source offsets of l_tile
    int l_tile = t_next, l_kt = 0;
    bool l_dead = false; // past the last tile of the list: the ring keeps turning on zero-fill loads
    unsigned a_voff[NA];
    [[maybe_unused]] int a_kq[NA];
    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int im_iy0[NCOL], im_ix0[NCOL], im_pix[NCOL];
    [[maybe_unused]] unsigned im_inv[NCOL];
    auto tile_origin = [&](int tile, int &m0, int &n0) {
        const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
        m0 = bm * BM;
        n0 = bn * BN;
    };
    auto setup_loader = [&](int tile) {
        int m0, n0;
        tile_origin(tile, m0, n0);
#pragma unroll
        for (int j = 0; j < NA; j++) {
            if constexpr (AL == A_M4) {
                const int f = (wave * NA + j) * 256 + lane * 4;
                const int k = f / BM, m = m0 + f % BM;
                a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
            } else {
                const int q = wave * NA + j, kq = q / (BM / 64), m = m0 + (q % (BM / 64)) * 64 + lane;
                a_kq[j] = kq * 4;
                a_voff[j] = m < p.M ? (unsigned)(((long long)m * p.a_rs + kq * 4) * 4) : OOB;
            }
        }
        if constexpr (BL == B_N4) {
#pragma unroll
            for (int j = 0; j < NBV; j++) {
                const int f = (wave * NBV + j) * 256 + lane * 4;
                const int k = f / BN, n = n0 + f % BN;
                const int nn = n < p.N ? n : 0;
                const int nb = nn / p.Pn, np = nn - nb * p.Pn;
                b_krow[j] = k;
                b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
            }
        } else {
#pragma unroll
            for (int c = 0; c < BN / 64; c++) {
                const int n = n0 + c * 64 + lane;
                const bool ok = n < p.N;
                const int nn = ok ? n : 0;
                const int nb = nn / p.Pn, np = nn - nb * p.Pn;
                const int oy = np / p.OW, ox = np - oy * p.OW;
                im_iy0[c] = ok ? oy * p.sy - p.pt : -0x40000000;
                im_ix0[c] = ox * p.sx - p.pl;
                im_pix[c] = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0[c];
                if constexpr (TAPS) {
                    unsigned colbad = 0;
                    for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0[c] + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
                    const unsigned allbad = (1u << p.KW) - 1u;
                    unsigned inv = 0x80000000u;
                    for (int ky = 0; ky < p.KH; ky++)
                        inv |= ((unsigned)(im_iy0[c] + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
                    im_inv[c] = inv;
                }
            }
        }
    };

    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    constexpr int LROWS = BK / 4;
    [[maybe_unused]] i32x2 lutE[LROWS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) { // LUT rows of the k-tile the loader issues NEXT (one table per conv geometry: tile independent)
        if constexpr (IM2COL) {
            const int krow0 = kt * BK + wave * LROWS;
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < LROWS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int stage) { // DMA of (l_tile, l_kt) into `stage`, then advance the loader
        float *As = smem + stage * STAGE;
        float *Bs = As + BK * BM;
        const int kt = l_kt;
        const unsigned a_soff = (unsigned)kt * a_kstep;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            bool dead = l_dead;
            if constexpr (AL == A_K4) dead = dead || a_kq[j] >= p.K - kt * BK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + (wave * NA + j) * 256), 16, (int)(dead ? OOB : a_voff[j]), (int)(l_dead ? 0u : a_soff), 0, 0);
        }
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * BK;
            const unsigned b_soff = (unsigned)kt * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBV + j) * 256), 16,
                                                         (int)((!l_dead && b_krow[j] < kleft) ? b_voff[j] : OOB), (int)(l_dead ? 0u : b_soff), 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NBG; j++) {
                constexpr int CPR = BN / 64;
                const int r = j / CPR, c = j % CPR;
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    voff = ((im_inv[c] << e[1]) & 0x80000000u) | ((unsigned)(im_pix[c] + e[0]) << 2);
                } else {
                    const int iy = im_iy0[c] + (e[1] & 0xffff);
                    const int ix = im_ix0[c] + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix[c] + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * LROWS + r) * BN + c * 64), 4, (int)(l_dead ? OOB : voff), 0, 0, 0);
            }
        }
        // advance: next k-tile of this tile, or the first k-tile of the next tile of the list (new per-lane offsets)
        if (!l_dead) {
            if (++l_kt == nk) {
                l_kt = 0;
                l_tile += t_step;
                if (l_tile < t_end) setup_loader(l_tile);
                else l_dead = true;
            }
        }
        fetch_lut(l_kt);
    };

    // ---- accumulators
    const int wq = t >> 6;
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
    f32x16 acc[MF16 ? 1 : TM][MF16 ? 1 : TN], tot[MF16 ? 1 : TM][MF16 ? 1 : TN];
    f32x4v acc4[MF16 ? TM2 : 1][MF16 ? TN2 : 1], tot4[MF16 ? TM2 : 1][MF16 ? TN2 : 1];
    auto zero_acc = [&]() {
        if constexpr (MF16) {
#pragma unroll
            for (int i = 0; i < TM2; i++)
#pragma unroll
                for (int j = 0; j < TN2; j++) acc4[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        }
    };

    auto compute_tile = [&](int stage) {
        if constexpr (MF16) {
            const float *As = smem + stage * STAGE + (AL == A_M4 ? wm0 + l15 + quad * BM : (wm0 + l15) * 4 + quad);
            auto a_idx = [](int ks, int i) { return AL == A_M4 ? 4 * ks * BM + i * 16 : ks * BM * 4 + i * 64; };
            const float *Bs = smem + stage * STAGE + BK * BM + wn0 + l15 + quad * BN;
            float af[2][TM2], bf[2][TN2];
#pragma unroll
            for (int i = 0; i < TM2; i++) af[0][i] = As[a_idx(0, i)];
#pragma unroll
            for (int j = 0; j < TN2; j++) bf[0][j] = Bs[j * 16];
#pragma unroll
            for (int ks = 0; ks < BK / 4; ks++) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < BK / 4) {
#pragma unroll
                    for (int i = 0; i < TM2; i++) af[nxt][i] = As[a_idx(ks + 1, i)];
#pragma unroll
                    for (int j = 0; j < TN2; j++) bf[nxt][j] = Bs[4 * (ks + 1) * BN + j * 16];
                }
#pragma unroll
                for (int i = 0; i < TM2; i++)
#pragma unroll
                    for (int j = 0; j < TN2; j++) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i], bf[cur][j], acc4[i][j], 0, 0, 0);
            }
        } else {
            const float *As = smem + stage * STAGE + (AL == A_M4 ? wm0 + l31 + half * BM : (wm0 + l31) * 4 + half);
            auto a_idx = [](int kk, int i) { return AL == A_M4 ? 2 * kk * BM + i * 32 : (kk >> 1) * BM * 4 + ((2 * kk) & 3) + i * 128; };
            const float *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
            if constexpr (MFK == 1) { // every fragment of the k-tile first, then the MFMAs with nothing between them
                float afa[BK / 2][TM], bfa[BK / 2][TN];
#pragma unroll
                for (int kk = 0; kk < BK / 2; kk++) {
#pragma unroll
                    for (int i = 0; i < TM; i++) afa[kk][i] = As[a_idx(kk, i)];
#pragma unroll
                    for (int j = 0; j < TN; j++) bfa[kk][j] = Bs[(2 * kk + half) * BN + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < BK / 2; kk++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afa[kk][i], bfa[kk][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            float af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[0][i] = As[a_idx(0, i)];
#pragma unroll
            for (int j = 0; j < TN; j++) bf[0][j] = Bs[half * BN + j * 32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < TM; i++) af[nxt][i] = As[a_idx(kk + 1, i)];
#pragma unroll
                    for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(2 * (kk + 1) + half) * BN + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_iglp_opt(0);
    };

    // fold of a finished depth block into `tot` / epilogue of a finished tile (the helpers of the one-tile kernels)
    auto flush = [&](bool first, int m0, int n0) {
        if constexpr (MF16) {
            int mb = m0 + wm0 + 4 * quad, nb0 = n0 + wn0 + l15;
            asm volatile("" : "+v"(mb), "+v"(nb0));
            if (first) fold_first16<TM2, TN2>(p, z, acc4, tot4, mb, nb0, c_zoff);
            else fold_next16<TM2, TN2>(p, acc4, tot4);
        } else {
            int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
            asm volatile("" : "+v"(mb), "+v"(nb0));
            if (first) fold_first<TM, TN>(p, z, acc, tot, mb, nb0, c_zoff);
            else fold_next<TM, TN>(p, acc, tot);
        }
        zero_acc();
    };
    auto finish = [&](bool single_block, int m0, int n0) {
        if constexpr (MF16) {
            int mb = m0 + wm0 + 4 * quad, nb0 = n0 + wn0 + l15;
            asm volatile("" : "+v"(mb), "+v"(nb0));
            if (single_block) {
                fold_first16<TM2, TN2>(p, z, acc4, acc4, mb, nb0, c_zoff);
                store_out16<TM2, TN2>(p, acc4, mb, nb0, c_zoff);
            } else {
                fold_next16<TM2, TN2>(p, acc4, tot4);
                store_out16<TM2, TN2>(p, tot4, mb, nb0, c_zoff);
            }
        } else {
            int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
            asm volatile("" : "+v"(mb), "+v"(nb0));
            if (single_block) {
                fold_first<TM, TN>(p, z, acc, acc, mb, nb0, c_zoff);
                store_out<TM, TN>(p, acc, mb, nb0, c_zoff);
            } else {
                fold_next<TM, TN>(p, acc, tot);
                store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
            }
        }
        zero_acc();
    };

    // ---- the ring: NSTAGE - 1 k-tiles in flight before the first MFMA, then one barrier per k-tile for the whole list
    setup_loader(l_tile);
    fetch_lut(0);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) issue_tile(i);
    zero_acc();
    int stage = 0;
    const bool single_block = nk <= KC_TILES;
#ifdef RTEN_TRACE
    unsigned long long tr_seg[5] = {0, 0, 0, 0, 0}, tr_n = 0, tr_prev = __builtin_readcyclecounter();
#define RTEN_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tr_seg[i] += now_ - tr_prev; tr_prev = now_; }
#else
#define RTEN_STAMP(i)
#endif
    for (int c_tile = t_next; c_tile < t_end; c_tile += t_step) {
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        for (int kt = 0; kt < nk; kt++) {
            RTEN_STAMP(4)
            // this wave's DMA for this k-tile has landed (younger loads: one more k-tile; stores of the previous tile's epilogue
            // can only make the count conservative: loads retire in order among themselves)
            if (!(ABLATE(p) & 32)) wait_vmcnt<PER_TILE *(NSTAGE - 2)>();
            RTEN_STAMP(0)
            if (!(ABLATE(p) & 8)) __builtin_amdgcn_s_barrier(); // ... and everyone else's; all waves are done reading the stage refilled next
            RTEN_STAMP(1)
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            if (!(ABLATE(p) & 1)) issue_tile(stp);
            RTEN_STAMP(2)
            if (ABLATE(p) & 16) { // ablation: the k-tile's MFMAs on register operands (no ds_read)
                if constexpr (!MF16) {
                    float fa = (float)kt, fb = (float)lane;
#pragma unroll
                    for (int kk = 0; kk < BK / 2; kk++)
#pragma unroll
                        for (int i = 0; i < TM; i++)
#pragma unroll
                            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i][j], 0, 0, 0);
                }
            } else if (!(ABLATE(p) & 2)) compute_tile(stage);
            RTEN_STAMP(3)
#ifdef RTEN_TRACE
            tr_n++;
#endif
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
            if (!single_block && kt + 1 < nk && (kt + 1) % KC_TILES == 0) flush(kt + 1 == KC_TILES, m0, n0);
        }
        finish(single_block, m0, n0);
    }
    wait_vmcnt<0>(); // the zero-fill look-ahead loads must land before the LDS goes away
#ifdef RTEN_TRACE
    if (blockIdx.x == 8 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 3))
        printf("[trace] wave %d: %llu k-tiles; cycles per k-tile: vmcnt wait %.0f, barrier %.0f, DMA issue + loader advance %.0f, fragments + MFMA issue %.0f, loop / fold / epilogue %.0f\n",
               wave, tr_n, (double)tr_seg[0] / tr_n, (double)tr_seg[1] / tr_n, (double)tr_seg[2] / tr_n, (double)tr_seg[3] / tr_n, (double)tr_seg[4] / tr_n);
#endif
#undef RTEN_STAMP
}

// =====================================================================================================
// Lean persistent kernel: 64x64 tiles, k-tiles of 32, prepacked (k-major) weights, dense or tap-masked im2col B.
//
// tools/probes/kloop.hip builds the k-loop of the kernels above piece by piece: 8 dependent MFMAs + their 16 LDS fragment reads
// + one barrier + the tile DMA cost a wave 666 cycles per k-tile (512 = matrix pipe) when NOTHING else is in the loop -- 120-130
// TFLOP/s with one or two workgroups per compute unit -- while the general kernels spend 1400: per-DMA selects for k-tails and
// dead tiles, LUT loads whose lgkmcnt(0) wait lands in the MFMA phase, depth-block / split / ablation branches, spilled scalars.
// This kernel is the probe's loop made real for the shapes that carry ResNet-50 (K a multiple of 32, alpha = 1, beta = 0):
//   * persistent workgroups walking an XCD-chunked tile list, DMA two k-tiles ahead across tile boundaries (as the kernel above);
//   * per k-tile and wave: 2 + 2 dwordx4 DMA (dense) or 2 + 8 dword gathers, 32 fragment reads, 16 MFMAs, ONE barrier; DMA source
//     offsets are plain per-tile registers (no selects: rows past K do not exist, rows past M / columns past N are out-of-range
//     offsets fixed at tile setup), the LUT rows of the k-tile after next are fetched by one s_load_dwordx16 AFTER the MFMAs are
//     issued, the tile-crossing bookkeeping sits in a cold branch;
//   * depth blocks of 256 = 8 k-tiles: the fold is one compare per k-tile.
// Numerics are those of every other kernel in this file (same chain per element): bit-identical.
// =====================================================================================================
constexpr int LBK = 32; // k-tile depth of the lean kernel

template <int BL, int NSTAGE = 3>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_f32_lean_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    static_assert(BL == B_N4 || BL == B_IM2COL_TAPS, "lean kernel: dense or tap-masked im2col B");
    constexpr int BM = 64, BN = 64;
    constexpr int STAGE = LBK * (BM + BN); // floats
    constexpr int NBG = LBK * BN / 64 / 4; // gather rows per wave per k-tile = 8
    constexpr int PER_TILE = 2 + (BL == B_N4 ? 2 : NBG);
    constexpr bool TAPS = BL == B_IM2COL_TAPS;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6;
    const int wm0 = (wq >> 1) * 32, wn0 = (wq & 1) * 32;

    const int T = p.tiles_m * p.tiles_n;
    int t_next, t_end, t_step;
    {
        const int id = blockIdx.x, G = (int)gridDim.x;
        const int xcd = id & 7, q = T >> 3, r = T & 7;
        const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_step = (G - xcd + 7) >> 3;
        t_next = lo + (id >> 3);
        t_end = lo + q + (xcd < r ? 1 : 0);
    }
    if (t_next >= t_end) return;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.B, 0, (int)p.b_bytes, 0x00020000);
    const int nk = p.K / LBK;
    const unsigned a_kstep = (unsigned)(LBK * p.a_cs * 4);
    const unsigned b_kstep = BL == B_N4 ? (unsigned)(LBK * p.b_rs * 4) : 0u;

    // ---- loader: (l_tile, l_kt) and the per-lane source offsets of l_tile (out-of-range lanes carry OOB: no select in the loop)
    int l_tile = t_next, l_kt = 0;
    unsigned a_voff[2], b_voff[2];
    [[maybe_unused]] int im_pix = 0;
    [[maybe_unused]] unsigned im_inv = 0;
    auto tile_origin = [&](int tile, int &m0, int &n0) __attribute__((always_inline)) {
        const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
        m0 = bm * BM;
        n0 = bn * BN;
    };
    auto setup_loader = [&](int tile) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(tile, m0, n0);
#pragma unroll
        for (int j = 0; j < 2; j++) { // DMA instruction q = 2 * wave + j moves k rows 4q .. 4q+3: lane -> row 4q + lane / 16, columns (lane % 16) * 4 ..
            const int q = wave * 2 + j, k = q * 4 + (lane >> 4), c = (lane & 15) * 4;
            const int m = m0 + c;
            a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
            if constexpr (BL == B_N4) {
                const int n = n0 + c;
                const int nn = n < p.N ? n : 0;
                const int nb = nn / p.Pn, np = nn - nb * p.Pn;
                b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
            }
        }
        if constexpr (TAPS) {
            const int n = n0 + lane; // gather instruction = one k row x 64 columns
            const bool ok = n < p.N;
            const int nn = ok ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            const int oy = np / p.OW, ox = np - oy * p.OW;
            const int iy0 = ok ? oy * p.sy - p.pt : -0x40000000, ix0 = ox * p.sx - p.pl;
            im_pix = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + ix0;
            unsigned colbad = 0;
            for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(ix0 + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
            const unsigned allbad = (1u << p.KW) - 1u;
            unsigned inv = 0x80000000u;
            for (int ky = 0; ky < p.KH; ky++) inv |= ((unsigned)(iy0 + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
            im_inv = inv;
        }
    };

    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    [[maybe_unused]] i32x2 lutE[NBG];
    [[maybe_unused]] auto fetch_lut = [&](int kt) __attribute__((always_inline)) { // LUT rows (8 consecutive entries = one s_load_dwordx16) of the k-tile the loader issues next
        if constexpr (TAPS) {
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut + (kt * LBK + wave * NBG);
#pragma unroll
            for (int j = 0; j < NBG; j++) lutE[j] = lc[j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        float *As = smem + stage * STAGE, *Bs = As + LBK * BM;
        const unsigned a_soff = (unsigned)l_kt * a_kstep;
#pragma unroll
        for (int j = 0; j < 2; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + (wave * 2 + j) * 256), 16, (int)a_voff[j], (int)a_soff, 0, 0);
        if constexpr (BL == B_N4) {
            const unsigned b_soff = (unsigned)l_kt * b_kstep;
#pragma unroll
            for (int j = 0; j < 2; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * 2 + j) * 256), 16, (int)b_voff[j], (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NBG; r++) {
                const i32x2 e = lutE[r];
                const unsigned voff = ((im_inv << e[1]) & 0x80000000u) | ((unsigned)(im_pix + e[0]) << 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBG + r) * BN), 4, (int)voff, 0, 0, 0);
            }
        }
    };
    auto advance_loader = [&]() __attribute__((always_inline)) { // next k-tile; crossing into the next tile of the list (or off its end: zero-fill loads) is the cold path
        if (__builtin_expect(++l_kt == nk, 0)) {
            l_kt = 0;
            l_tile += t_step;
            if (l_tile < t_end) {
                setup_loader(l_tile);
            } else {
                a_voff[0] = a_voff[1] = b_voff[0] = b_voff[1] = OOB;
                if constexpr (TAPS) im_inv = 0xffffffffu; // every tap reads out of range
            }
        }
    };

    f32x16 acc[1][1], tot[1][1];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;

    auto compute_tile = [&](int stage) __attribute__((always_inline)) {
        const float *As = smem + stage * STAGE + wm0 + l31 + half * BM;
        const float *Bs = smem + stage * STAGE + LBK * BM + wn0 + l31 + half * BN;
        float af[2], bf[2];
        af[0] = As[0];
        bf[0] = Bs[0];
#pragma unroll
        for (int kk = 0; kk < LBK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < LBK / 2) {
                af[nxt] = As[2 * (kk + 1) * BM];
                bf[nxt] = Bs[2 * (kk + 1) * BN];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc[0][0], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
    };

    setup_loader(l_tile);
    fetch_lut(0);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) {
        issue_tile(i);
        advance_loader();
        fetch_lut(l_kt);
    }
    // De-phase the workgroups that share a compute unit: they run identical tile lists and would otherwise reach their
    // epilogues (and tile-crossing setup) together, leaving the matrix pipe idle for both.  Workgroup ids are dealt round-robin
    // over the compute units, so id / num_cus is the residency slot; slot s waits s / slots of a tile's matrix time once.
    if (p.debug & 0x100) {
        const int slots = (int)gridDim.x / p.split_slots, slot = slots > 1 ? (int)blockIdx.x / p.split_slots : 0; // split_slots = num_cus here
        if (slot > 0) {
            const int ticks = nk * 16 * 64 * slot / slots / 64; // s_sleep unit = 64 cycles
            for (int i = 0; i < ticks; i += 8) __builtin_amdgcn_s_sleep(8);
        }
    }
    int stage = 0;
    const bool single_block = nk <= 256 / LBK;
    for (int c_tile = t_next; c_tile < t_end; c_tile += t_step) {
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        for (int kt = 0; kt < nk; kt++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>(); // this wave's DMA for this k-tile has landed (NSTAGE - 2 younger k-tiles stay in flight)
            __builtin_amdgcn_s_barrier();
            issue_tile(stage == 0 ? NSTAGE - 1 : stage - 1);
            compute_tile(stage);
            advance_loader();
            fetch_lut(l_kt); // consumed by the NEXT iteration's issue: the scalar load's latency runs under this k-tile's MFMAs
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
            if (__builtin_expect(((kt + 1) & 7) == 0 && kt + 1 < nk, 0)) { // depth-block boundary (256 = 8 k-tiles)
                int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
                asm volatile("" : "+v"(mb), "+v"(nb0));
                if (kt + 1 == 8) fold_first<1, 1>(p, 0, acc, tot, mb, nb0, 0);
                else fold_next<1, 1>(p, acc, tot);
#pragma unroll
                for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;
            }
        }
        {
            int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
            asm volatile("" : "+v"(mb), "+v"(nb0));
            if (single_block) {
                fold_first<1, 1>(p, 0, acc, acc, mb, nb0, 0);
                store_out<1, 1>(p, acc, mb, nb0, 0);
            } else {
                fold_next<1, 1>(p, acc, tot);
                store_out<1, 1>(p, tot, mb, nb0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;
        }
    }
    wait_vmcnt<0>();
}

template <int BM, int BN, int BL, bool MULTI_KC>
__global__ __launch_bounds__(2 * NTHREADS, 2) void igemm_f32_ws_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "DMA kernel covers the conv operand layouts");
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STAGE = BK * (BM + BN); // floats per stage
    constexpr int NA = BK * BM / 256 / 4; // dwordx4 DMA instructions per wave per tile (A)
    constexpr int NBV = BK * BN / 256 / 4; // dwordx4 (dense B)
    constexpr int NBG = BK * BN / 64 / 4;  // dword gathers per wave per tile (im2col B)
    constexpr int PER_TILE = NA + (BL == B_N4 ? NBV : NBG);
    static_assert(NA >= 1 && NBV >= 1, "tile too small for 4-wave DMA split");
    constexpr int NSTAGE = nstage_for(BM, BN);
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    // 8 waves: 0..3 multiply (one per SIMD), 4..7 are loader waves that only issue LDS-DMA.  The two roles share
    // each SIMD, so address arithmetic / DMA issue of the loader overlaps the MFMA wave's matrix-pipe time even
    // when this is the only workgroup on the CU.
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool is_loader = wave_all >= 4;
    const int wave = wave_all & 3; // loader index or MFMA wave index
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;

    int tile;
    {
        const int nt = p.tiles_m * p.tiles_n;
        const int id = blockIdx.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + BK - 1) / BK;

    // ---- loop-invariant DMA source offsets.  Wave w issues instructions q = w*N + j; instruction q covers
    // the flat tile range [q*256, q*256+256) floats (dwordx4) or [q*64, q*64+64) (dword gather).
    unsigned a_voff[NA];
#pragma unroll
    for (int j = 0; j < NA; j++) {
        const int f = (wave * NA + j) * 256 + lane * 4;
        const int k = f / BM, m = m0 + f % BM;
        // rows >= K lie past the end of the [K][M4] buffer (hardware range check); columns >= M4 must not wrap
        a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
    }
    const unsigned a_kstep = (unsigned)(BK * p.a_cs * 4);

    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] unsigned b_kstep = 0;
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    constexpr int NCOL = IM2COL && BN == 128 ? 2 : 1;
    [[maybe_unused]] int im_iy0[NCOL], im_ix0[NCOL], im_pix[NCOL];
    [[maybe_unused]] unsigned im_inv[NCOL]; // TAPS: bit t set = tap t of this lane's pixel is padding; bit 31 always set (k-tail rows)
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NBV; j++) {
            const int f = (wave * NBV + j) * 256 + lane * 4;
            const int k = f / BN, n = n0 + f % BN;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(BK * p.b_rs * 4);
    } else {
        // gather instruction q = wave*NBG + j covers row q / (BN/64), columns (q % (BN/64))*64 + lane.
        // A lane therefore sees at most BN/64 distinct columns.
#pragma unroll
        for (int c = 0; c < BN / 64; c++) {
            const int n = n0 + c * 64 + lane;
            const bool ok = n < p.N;
            const int nn = ok ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            const int oy = np / p.OW, ox = np - oy * p.OW;
            im_iy0[c] = ok ? oy * p.sy - p.pt : -0x40000000;
            im_ix0[c] = ox * p.sx - p.pl;
            im_pix[c] = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0[c];
            if constexpr (TAPS) {
                unsigned colbad = 0; // bit kx set: column tap kx falls outside the image
                for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0[c] + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
                const unsigned allbad = (1u << p.KW) - 1u;
                unsigned inv = 0x80000000u;
                for (int ky = 0; ky < p.KH; ky++)
                    inv |= ((unsigned)(im_iy0[c] + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
                im_inv[c] = inv;
            }
        }
    }

    // im2col LUT entries (scalar loads) for the tile whose DMA is issued NEXT
    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    constexpr int LROWS = BK / 4; // rows of a tile handled by one wave (NBG / (BN/64))
    [[maybe_unused]] i32x2 lutE[LROWS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (IM2COL) {
            const int krow0 = kt * BK + wave * LROWS;
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < LROWS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + BK * BM;
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0); // keep the scalar offset inside the buffer
        const bool past = kt >= nk;
        const unsigned a_soff = (unsigned)kts * a_kstep;
#pragma unroll
        for (int j = 0; j < NA; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + (wave * NA + j) * 256), 16,
                                                     (int)(past ? OOB : a_voff[j]), (int)a_soff, 0, 0);
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * BK;
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBV + j) * 256), 16,
                                                         (int)(b_krow[j] < kleft ? b_voff[j] : OOB), (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NBG; j++) {
                constexpr int CPR = BN / 64;            // gather instructions per tile row
                const int r = j / CPR, c = j % CPR;     // row within this wave's LROWS, column chunk
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    // e[1] = 31 - tap: the tap's padding bit moves to bit 31 and pushes the offset out of range
                    voff = ((im_inv[c] << e[1]) & 0x80000000u) | ((unsigned)(im_pix[c] + e[0]) << 2);
                } else {
                    const int iy = im_iy0[c] + (e[1] & 0xffff);
                    const int ix = im_ix0[c] + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix[c] + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * LROWS + r) * BN + c * 64), 4, (int)voff, 0, 0, 0);
            }
        }
    };

    // ---- accumulators / epilogue helpers (same numerics as igemm_f32_kernel)
    const int wq = (t >> 6) & 3; // per-lane copy of the MFMA wave id for address math
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    f32x16 acc[TM][TN];
    [[maybe_unused]] f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    [[maybe_unused]] auto flush = [&](bool first) {
        int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
        asm volatile("" : "+v"(mb), "+v"(nb0));
        if (first) fold_first<TM, TN>(p, z, acc, tot, mb, nb0, c_zoff);
        else fold_next<TM, TN>(p, acc, tot);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };

    auto compute_tile = [&](int stage) {
        const float *As = smem + stage * STAGE + wm0 + l31;
        const float *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
        float af[2][TM], bf[2][TN]; // operand fragments, double buffered across k-pairs
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = As[half * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = Bs[half * BN + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[nxt][i] = As[(2 * (kk + 1) + half) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(2 * (kk + 1) + half) * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0); // interleave the next group's ds_reads behind the current group's first MFMA
    };

    // ---- software pipeline: tiles kt+1 and kt+2 are in flight while tile kt is multiplied.  One s_barrier per
    // k-tile, executed by all 8 waves: loaders arrive after their DMA of tile kt has landed, MFMA waves after they
    // finished tile kt-1; past the barrier the loaders refill the freed stage while the MFMA waves multiply.
    const int nblk = MULTI_KC ? (nk + KC_TILES - 1) / KC_TILES : 1;
    if (is_loader) {
        fetch_lut(0);
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; i++) {
            issue_tile(i, i);
            fetch_lut(i + 1);
        }
        int stage = 0;
        for (int kt = 0; kt < nk; kt++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>();
            __builtin_amdgcn_s_barrier();
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            if (!(ABLATE(p) & 1)) issue_tile(kt + NSTAGE - 1, stp);
            fetch_lut(kt + NSTAGE);
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        wait_vmcnt<0>(); // the look-ahead tiles (out of range, zero fill) must land before the LDS goes away
        return;
    }
    {
        int stage = 0;
        for (int blk = 0; blk < nblk; blk++) {
            const int kt_end = MULTI_KC ? ((blk + 1) * KC_TILES < nk ? (blk + 1) * KC_TILES : nk) : nk;
            for (int kt = blk * KC_TILES; kt < kt_end; kt++) {
                __builtin_amdgcn_s_barrier();
                if (!(ABLATE(p) & 2)) compute_tile(stage);
                stage = stage == NSTAGE - 1 ? 0 : stage + 1;
            }
            if constexpr (MULTI_KC) {
                if (blk + 1 < nblk) flush(blk == 0);
            }
        }
    }

    if (!(ABLATE(p) & 4)) {
        const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
        if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
            fold_next<TM, TN>(p, acc, tot);
            store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
        } else {
            fold_first<TM, TN>(p, z, acc, acc, mb, nb0, c_zoff);
            store_out<TM, TN>(p, acc, mb, nb0, c_zoff);
        }
    }
}

// =====================================================================================================
// Thin-tile tail kernel: 16 (m) x 64 (n) tiles on v_mfma_f32_16x16x4_f32.
//
// The f32 matrix pipe makes a 32x32 accumulator block x full K a long indivisible unit on one SIMD, and ResNet's column
// counts (batch x 49 x 2^k) leave a fraction of a round of 64x64 tiles over: the chip then idles 12-25 % of the layer's
// time behind a few straggler tiles.  A launch plan with `split_mode == 4` gives the whole rounds to the 64x64 kernel
// (columns [0, n_lo)) and the remaining columns to this kernel, whose waves own 16x16 blocks -- a quarter of the work per
// SIMD, so the tail costs a quarter of a round and every CU takes part.
// v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain like the 32x32x2 form (tools/probes/mfma_16x16x4_order.hip), so an
// output element sees the same chain: depth blocks of 256 folded with separate adds, bias after the first block --
// bit-identical to the big tiles and to the reference.
// Tile DMA as in the kernel above (A k-major [K][M4] -> LDS [32][32] with rows >= 16 zero filled, B dense or im2col
// gather -> LDS [32][64]), three stages, k-tiles of 32.  alpha == 1, beta == 0 (convolution) only.
// =====================================================================================================
typedef float f32x4acc __attribute__((ext_vector_type(4)));
constexpr int TBK = 32;               // k-tile depth of the thin kernel
constexpr int TKC_TILES = 256 / TBK;  // k-tiles per reference depth block

template <int BL, bool MULTI_KC>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_f32_thin_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "thin kernel covers the conv operand layouts");
    constexpr int BM = 16, BN = 64, LDA = 32;        // LDS A image is 32 wide (one dwordx4 DMA instruction per wave), 16 used
    constexpr int STAGE = TBK * (LDA + BN);          // floats per stage
    constexpr int NBV = TBK * BN / 256 / 4;          // dwordx4 per wave per tile (dense B) = 2
    constexpr int NBG = TBK * BN / 64 / 4;           // dword gathers per wave per tile (im2col B) = 8
    constexpr int PER_TILE = 1 + (BL == B_N4 ? NBV : NBG);
    constexpr int NSTAGE = 3;
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    const int z = blockIdx.y;
    const int tile = blockIdx.x;
    const int bm = tile % p.tiles_m, bn = tile / p.tiles_m; // m fastest: the workgroups of one column strip share the B panel in L2
    const int m0 = bm * BM, n0 = p.n_lo + bn * BN;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + TBK - 1) / TBK;

    // A: wave w moves rows [8w, 8w+8) of the k-tile: lane -> (row 8w + lane/8, columns (lane%8)*4 ..+3 of the 32-wide LDS image)
    unsigned a_voff;
    {
        const int k = wave * 8 + (lane >> 3), ml = (lane & 7) * 4, m = m0 + ml;
        a_voff = (ml < BM && m < (int)p.a_cs) ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
    }
    const unsigned a_kstep = (unsigned)(TBK * p.a_cs * 4);

    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] unsigned b_kstep = 0;
    [[maybe_unused]] int im_iy0 = 0, im_ix0 = 0, im_pix = 0;
    [[maybe_unused]] unsigned im_inv = 0;
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NBV; j++) {
            const int f = (wave * NBV + j) * 256 + lane * 4;
            const int k = f / BN, n = n0 + f % BN;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(TBK * p.b_rs * 4);
    } else {
        const int n = n0 + lane; // gather instruction = one k row x 64 columns: a lane sees one column
        const bool ok = n < p.N;
        const int nn = ok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const int oy = np / p.OW, ox = np - oy * p.OW;
        im_iy0 = ok ? oy * p.sy - p.pt : -0x40000000;
        im_ix0 = ox * p.sx - p.pl;
        im_pix = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0;
        if constexpr (TAPS) {
            unsigned colbad = 0;
            for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0 + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
            const unsigned allbad = (1u << p.KW) - 1u;
            unsigned inv = 0x80000000u;
            for (int ky = 0; ky < p.KH; ky++) inv |= ((unsigned)(im_iy0 + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
            im_inv = inv;
        }
    }

    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    constexpr int LROWS = TBK / 4; // rows of a k-tile gathered by one wave
    [[maybe_unused]] i32x2 lutE[LROWS];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (IM2COL) {
            const int krow0 = kt * TBK + wave * LROWS;
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < LROWS; j++) lutE[j] = lc[krow0 + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + TBK * LDA;
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0); // keep the scalar offset inside the buffer
        const bool past = kt >= nk;
        // rows >= K lie past the end of the [K][M4] buffer: the hardware range check zero-fills them
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + wave * 256), 16, (int)(past ? OOB : a_voff), (int)((unsigned)kts * a_kstep), 0, 0);
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * TBK;
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * NBV + j) * 256), 16, (int)(b_krow[j] < kleft ? b_voff[j] : OOB), (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NBG; r++) {
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    voff = ((im_inv << e[1]) & 0x80000000u) | ((unsigned)(im_pix + e[0]) << 2);
                } else {
                    const int iy = im_iy0 + (e[1] & 0xffff);
                    const int ix = im_ix0 + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + (wave * LROWS + r) * BN), 4, (int)voff, 0, 0, 0);
            }
        }
    };

    const int wq = t >> 6;       // per-lane copy of the wave id for address math
    const int wn0 = wq * 16;     // the four waves own 16-column strips of the 64-column tile
    f32x4acc acc = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] f32x4acc tot = {0.f, 0.f, 0.f, 0.f};

    auto compute_tile = [&](int stage) {
        // fragments of k-step kk (4 rows of the tile): A[m = l15][k = 4 kk + quad], B[k = 4 kk + quad][n = wn0 + l15]
        const float *As = smem + stage * STAGE + quad * LDA + l15;
        const float *Bs = smem + stage * STAGE + TBK * LDA + quad * BN + wn0 + l15;
        float af[TBK / 4], bf[TBK / 4];
#pragma unroll
        for (int kk = 0; kk < TBK / 4; kk++) {
            af[kk] = As[kk * 4 * LDA];
            bf[kk] = Bs[kk * 4 * BN];
        }
#pragma unroll
        for (int kk = 0; kk < TBK / 4; kk++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bf[kk], acc, 0, 0, 0);
    };

    // element r of the accumulator: row m0 + 4 * quad + r, column n0 + wn0 + l15
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void *)((p.bias ? p.bias : p.C) + (long long)z * p.bias_bs), 0, 0x7ffffffc, 0x00020000);
    auto first_block = [&](f32x4acc a) { // alpha == 1, beta == 0: out = acc, then the bias (rten-gemm/src/lib.rs:1008-1013,1221-1255)
        f32x4acc v = a;
        if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) {
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 4 * quad + r;
                b4[r] = buf_load1(rsBias, m < p.M ? (unsigned)m << 2 : OOB, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = v[r] + b4[r];
        }
        return v;
    };

    const int nblk = MULTI_KC ? (nk + TKC_TILES - 1) / TKC_TILES : 1;
    fetch_lut(0);
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) {
        issue_tile(i, i);
        fetch_lut(i + 1);
    }
    int stage = 0;
    for (int blk = 0; blk < nblk; blk++) {
        const int kt_end = MULTI_KC ? ((blk + 1) * TKC_TILES < nk ? (blk + 1) * TKC_TILES : nk) : nk;
        for (int kt = blk * TKC_TILES; kt < kt_end; kt++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>();
            __builtin_amdgcn_s_barrier();
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            issue_tile(kt + NSTAGE - 1, stp);
            fetch_lut(kt + NSTAGE);
            compute_tile(stage);
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        if constexpr (MULTI_KC) {
            if (blk + 1 < nblk) {
                if (blk == 0) tot = first_block(acc);
                else {
#pragma unroll
                    for (int r = 0; r < 4; r++) tot[r] = tot[r] + acc[r];
                }
                acc = f32x4acc{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    wait_vmcnt<0>(); // drain the look-ahead tiles before the LDS goes away

    f32x4acc v;
    if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = tot[r] + acc[r];
    } else {
        v = first_block(acc);
    }
    // residual Add, activation, NCHW / row-major store
    const int n = n0 + wn0 + l15;
    const bool cok = n < p.N;
    const int nn = cok ? n : 0;
    const int nb = nn / p.Pn, np = nn - nb * p.Pn;
    const unsigned col = (unsigned)((long long)nb * p.c_ns + np);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + c_zoff), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)((p.res ? p.res : p.C) + c_zoff), 0, 0x7ffffffc, 0x00020000);
    unsigned voff[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = m0 + 4 * quad + r;
        voff[r] = (cok && m < p.M) ? (col + (unsigned)m * (unsigned)p.c_rs) << 2 : OOB;
    }
    if (p.res != nullptr) {
        float rr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) rr[r] = buf_load1(rsR, voff[r], 0);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = v[r] + rr[r];
    }
    if (p.act == RTEN_HIP_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = vm::relu(v[r]);
    } else if (p.act == RTEN_HIP_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = vm::gelu(v[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float x = v[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rsC, (int)voff[r], 0, 0);
    }
}

// im2col lookup table: entry k -> {c*HW + ky*dy*W + kx*dx, (ky*dy) | (kx*dx) << 16}; rows >= K get an
// offset pair that fails every bounds test.  taps != 0 (B_IM2COL_TAPS): the second word is 31 - (ky*KW + kx), the
// left shift that moves the tap's padding bit of the per-lane mask to bit 31; rows >= K use shift 0 (bit 31 is
// always set in the masks).  Built once per conv geometry and cached in the context.
__global__ void im2col_lut_kernel(i32x2 *lut, int K, int Kpad, int KHW, int KW, int HW, int W, int dy, int dx, int taps) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kpad) return;
    if (k >= K) { lut[k] = taps ? i32x2{0, 0} : i32x2{0, 0xffff}; return; } // dy = 65535 > any padded height
    const int c = k / KHW, rem = k - c * KHW, ky = rem / KW, kx = rem - ky * KW;
    lut[k] = i32x2{c * HW + ky * dy * W + kx * dx, taps ? 31 - rem : (ky * dy) | ((kx * dx) << 16)};
}

// =====================================================================================================
// Small-M weight streaming (variant 31; the automatic choice for M <= 64, one batch): the classifier / projection shape, where the
// tiled kernels have a handful of 64-row tiles and one of them streams megabytes of B alone (ResNet-50's 32 x 2048 x 1000 Gemm:
// 16 workgroups, 19-25 us for an 8 MB weight read).  Here a WAVE owns one 16x16 block of C for ONE depth block (kc = 256): a chain of
// 64 dependent v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain, tools/probes/mfma_16x16x4_order.hip) -- the reference's micro-kernel
// chain for that block (rten-gemm/src/kernels/simd_generic.rs:326-367) -- and a workgroup = 4 such waves sharing 16 * MT rows of A and
// 64 / MT * 16 columns of B through LDS.  grid = column groups x depth blocks, so ceil(N / (64 / MT * 16)) * ceil(K / 256) workgroups stream B
// (classifier: 32 x 8 = 256, one per compute unit, 64 KB each).  The raw block sums go to the split-K slab; the last workgroup of
// a 16x16 block to arrive folds them in depth-block order (first block: beta * C + bias; later blocks: separate adds;
// rten-gemm/src/lib.rs:1008-1013,1221-1255) -- same bits as the unsplit chain, same visibility protocol as split_finish.
//
// LDS image of an operand row (an A row or a B column; 256 depths = 1 KB): depth k = 16 j + 4 i + g stored at 16 j + 4 g + i, the 16-byte groups of a
// row XOR-swizzled by the row (sm_at): lane (row = lane % 16, g = lane / 16) reads the operands of its MFMAs 4 j .. 4 j + 3 with ONE conflict-free
// ds_read_b128, a loader lane holding depths 16 j + 4 q + {0..3} of a row writes four words (conflict free), and 64 or 80 rows are 64 / 80 KB:
// TWO workgroups per compute unit, so one's operand fetch runs under the other's MFMA chain.
// =====================================================================================================
constexpr int SM_LD = 256;
// float index of (operand row, position in the row's image): 16-byte groups XOR-swizzled by the row so that 16 rows at one position fall on 16 different bank groups
__device__ __forceinline__ int sm_at(int row, int pos) { return row * SM_LD + (pos ^ ((row & 15) << 2)); }

template <int MT>
__global__ __launch_bounds__(256) void gemm_f32_smallm_kernel(const GemmArgs p) {
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    constexpr int NW = 4 / MT;               // 16-column blocks per workgroup
    constexpr int RA = 16 * MT, RB = 16 * NW; // operand rows in LDS
    __shared__ __attribute__((aligned(16))) float smem[(RA + RB) * SM_LD];
    float *As = smem, *Bs = smem + RA * SM_LD;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    const int gx = blockIdx.x, kb = blockIdx.y;
    const int n0 = gx * RB, k0 = kb * 256;
    const int depth = p.K - k0 < 256 ? p.K - k0 : 256;
    const int nblk = (int)gridDim.y;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.B, 0, (int)p.b_bytes, 0x00020000);

    // ---- stage the operands: depth-contiguous rows take 16-byte loads (a wave covers 16 rows x 64 bytes per depth group j)
    const bool a_vec = p.tiles_m & 1, b_vec = p.tiles_m & 2; // set by the launcher: depth stride 1, row stride and K multiples of 4, base 16-byte aligned
    if (a_vec) {
        f32x4 v[MT * 4];
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int f = u * 256 + t, q = f & 3, r = (f >> 2) & 15, j = (f >> 6) & 15, rg = f >> 10;
            const int row = rg * 16 + r, k = 16 * j + 4 * q;
            v[u] = buf_load4(rsA, (row < p.M && k < depth) ? (unsigned)(((long long)row * p.a_rs + k0 + k) << 2) : OOB, 0);
        }
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int f = u * 256 + t, q = f & 3, r = (f >> 2) & 15, j = (f >> 6) & 15, rg = f >> 10;
            const int row = rg * 16 + r;
            As[sm_at(row, 16 * j + q)] = v[u][0]; As[sm_at(row, 16 * j + 4 + q)] = v[u][1]; As[sm_at(row, 16 * j + 8 + q)] = v[u][2]; As[sm_at(row, 16 * j + 12 + q)] = v[u][3];
        }
    } else if (p.tiles_m & 4) { // A given transposed ([K][M], 16-byte groups of rows): a wave covers 16 depths x 16 rows per load
        f32x4 v[MT * 4];
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int f = u * 256 + t, c4 = f & 3, k = 16 * ((f >> 6) & 15) + ((f >> 2) & 15), row = 16 * (f >> 10) + 4 * c4;
            v[u] = buf_load4(rsA, (row < p.M && k < depth) ? (unsigned)(((long long)(k0 + k) * p.a_cs + row) << 2) : OOB, 0);
        }
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int f = u * 256 + t, c4 = f & 3, k = 16 * ((f >> 6) & 15) + ((f >> 2) & 15), row = 16 * (f >> 10) + 4 * c4;
            const int pos = (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3);
            As[sm_at(row, pos)] = v[u][0]; As[sm_at(row + 1, pos)] = v[u][1]; As[sm_at(row + 2, pos)] = v[u][2]; As[sm_at(row + 3, pos)] = v[u][3];
        }
    } else {
        const bool along_m = p.a_dir_m; // rows are the contiguous direction (A given transposed): lanes walk rows
#pragma unroll 8
        for (int u = 0; u < RA; u++) {
            const int idx = u * 256 + t;
            const int row = along_m ? idx % RA : idx >> 8, k = along_m ? idx / RA : idx & 255;
            const float x = buf_load1(rsA, (row < p.M && k < depth) ? (unsigned)(((long long)row * p.a_rs + (long long)(k0 + k) * p.a_cs) << 2) : OOB, 0);
            As[sm_at(row, (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3))] = x;
        }
    }
    if (b_vec) {
        f32x4 v[NW * 4];
#pragma unroll
        for (int u = 0; u < NW * 4; u++) {
            const int f = u * 256 + t, q = f & 3, r = (f >> 2) & 15, j = (f >> 6) & 15, rg = f >> 10;
            const int col = n0 + rg * 16 + r, k = 16 * j + 4 * q;
            v[u] = buf_load4(rsB, (col < p.N && k < depth) ? (unsigned)(((long long)col * p.b_cs + k0 + k) << 2) : OOB, 0);
        }
#pragma unroll
        for (int u = 0; u < NW * 4; u++) {
            const int f = u * 256 + t, q = f & 3, r = (f >> 2) & 15, j = (f >> 6) & 15, rg = f >> 10;
            const int row = rg * 16 + r;
            Bs[sm_at(row, 16 * j + q)] = v[u][0]; Bs[sm_at(row, 16 * j + 4 + q)] = v[u][1]; Bs[sm_at(row, 16 * j + 8 + q)] = v[u][2]; Bs[sm_at(row, 16 * j + 12 + q)] = v[u][3];
        }
    } else if (p.tiles_m & 8) { // B as [K][N] (16-byte groups of columns): a wave covers 16 depths x 16 columns per load
        f32x4 v[NW * 4];
#pragma unroll
        for (int u = 0; u < NW * 4; u++) {
            const int f = u * 256 + t, c4 = f & 3, k = 16 * ((f >> 6) & 15) + ((f >> 2) & 15), c = 16 * (f >> 10) + 4 * c4;
            v[u] = buf_load4(rsB, (n0 + c < p.N && k < depth) ? (unsigned)(((long long)(k0 + k) * p.b_rs + n0 + c) << 2) : OOB, 0);
        }
#pragma unroll
        for (int u = 0; u < NW * 4; u++) {
            const int f = u * 256 + t, c4 = f & 3, k = 16 * ((f >> 6) & 15) + ((f >> 2) & 15), c = 16 * (f >> 10) + 4 * c4;
            const int pos = (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3);
            Bs[sm_at(c, pos)] = v[u][0]; Bs[sm_at(c + 1, pos)] = v[u][1]; Bs[sm_at(c + 2, pos)] = v[u][2]; Bs[sm_at(c + 3, pos)] = v[u][3];
        }
    } else {
        const bool along_n = p.b_dir_n; // columns are the contiguous direction (B as [K][N]): lanes walk columns
#pragma unroll 8
        for (int u = 0; u < RB; u++) {
            const int idx = u * 256 + t;
            const int c = along_n ? idx % RB : idx >> 8, k = along_n ? idx / RB : idx & 255;
            const int col = n0 + c;
            const float x = buf_load1(rsB, (col < p.N && k < depth) ? (unsigned)(((long long)(k0 + k) * p.b_rs + (long long)col * p.b_cs) << 2) : OOB, 0);
            Bs[sm_at(c, (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3))] = x;
        }
    }
    __syncthreads();

    // ---- one 16x16 block of C, one depth block: the MFMA chain
    const int mt = wave % MT, nw = wave / MT;
    f32x4 af[16], bf[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        af[j] = *(const f32x4 *)(As + sm_at(mt * 16 + l15, 16 * j + 4 * quad));
        bf[j] = *(const f32x4 *)(Bs + sm_at(nw * 16 + l15, 16 * j + 4 * quad));
    }
    f32x4v acc[1][1];
    acc[0][0] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (16 * j + 4 * i < depth) // (uniform; depths past the end inside the last MFMA are zeros: exact)
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][i], bf[j][i], acc[0][0], 0, 0, 0);
        }
    }

    const int mb = mt * 16 + 4 * quad, nb0 = n0 + nw * 16 + l15;
    if (nblk == 1) {
        fold_first16<1, 1>(p, 0, acc, acc, mb, nb0, 0);
        store_out16<1, 1>(p, acc, mb, nb0, 0);
        return;
    }
    // ---- park the raw sums; the last arrival of the column group folds them in depth-block order
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.slab + (long long)gx * nblk * 1024), 0, nblk * 4096, 0x00020000);
    coherent_store4(rs, (unsigned)((kb * 4 + wave) * 64 + lane) * 16u, acc[0][0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through store has been acknowledged
    // per-WAVE arrival (counter gx * 4 + wave): the nblk waves that own the same 16x16 block meet on it, no workgroup barrier on the way
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(p.split_counters + gx * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)nblk - 1u) return;
    // all of a batch's slots are in flight before the first fold (one memory round trip per 8 depth blocks, not one per block)
    f32x4v tot[1][1], cur[1][1];
    f32x4 part[8];
    for (int s0 = 0; s0 < nblk; s0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) part[u] = coherent_load4(rs, s0 + u < nblk ? (unsigned)(((s0 + u) * 4 + wave) * 64 + lane) * 16u : OOB);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (s0 + u < nblk) {
                cur[0][0] = part[u];
                if (s0 + u == 0) fold_first16<1, 1>(p, 0, cur, tot, mb, nb0, 0);
                else fold_next16<1, 1>(p, cur, tot);
            }
        }
    }
    store_out16<1, 1>(p, tot, mb, nb0, 0);
    if (lane == 0) p.split_counters[gx * 4 + wave] = 0u;
}

} // namespace

// =====================================================================================================
// Host side: variant selection and the C-ABI entry points
// =====================================================================================================
namespace {

struct TileCfg { int bm, bn; float penalty; };
// variant ids are part of the tuning interface (rten_hip_set_gemm_variant_override)
constexpr TileCfg kCfgs[4] = {{128, 128, 1.00f}, {128, 64, 1.04f}, {64, 128, 1.04f}, {64, 64, 1.12f}};

// Occupancy cap of the LDS-DMA kernels (order bits 4-6 = workgroups per compute unit, 0 = whatever fits): dynamic LDS bytes the kernel never
// touches, sized so that exactly `cap` workgroups fit the 160 KB of a compute unit.  Why a cap: with one 32x32 accumulator block per wave
// every MFMA of a wave depends on its previous one, and three or more such waves on a SIMD share the matrix pipe badly (tools/debug/f32_trace.py:
// 64x64 tiles at 3-4 workgroups per CU keep it 67 % busy inside the k-loop; tools/probes/kloop.hip row A: 150 TF/s at two waves per SIMD, 114 at three).
inline size_t occupancy_pad(const rten_hip_ctx *ctx, int static_bytes) {
    static const int env_cap = getenv("RTEN_HIP_OCC_CAP") ? atoi(getenv("RTEN_HIP_OCC_CAP")) : 0; // tuning only
    const int cap = env_cap > 0 ? env_cap : (ctx->tile_order >> 4) & 7;
    if (cap < 2) return 0; // (a cap of 1 needs more than 64 KB per workgroup: not offered)
    const int want = 160 * 1024 / (cap + 1) + 1024; // one byte more than what cap + 1 workgroups could share
    return want > static_bytes ? (size_t)((want - static_bytes + 255) & ~255) : 0;
}

// One launch plan for every pipeline: whole tiles [0, t1) by the folding kernel (MODE 0/1), split tiles [t1, T) by
// the split-K producer (MODE 2) followed by the ordered fixup.  Only depth-block boundaries are legal K cuts.
template <int BM, int BN, int AL, int BL>
int32_t launch_cfg(rten_hip_ctx *ctx, GemmArgs &a, int Z) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const double flops = 2.0 * a.M * (double)a.N * a.K * Z;
    const double bytes = 4.0 * Z * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N);
    char kname[96];
    const bool multi = a.K > 256;
    constexpr bool kDma = (AL == A_M4 || AL == A_K4) && (BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS);
    int pipe = kDma ? ctx->pipeline : 0;
    if (AL == A_K4 && pipe == 2) pipe = 1; // the wave-specialised kernel only takes k-major A
    if (pipe == 6 && !(AL == A_M4 && BM == 64 && BN == 64)) pipe = 1; // the wave-tile kernels: prepacked weights, 64x64 tiles
    if (pipe == 6 && ctx->wave_flavour >= 4 && BL != B_N4) pipe = 1;   // 32x32 wave tiles: dense B only
    if (pipe == 7 && !(BM == 64 && BN == 64)) pipe = 1;                // the two-stage ring exists for 64x64 tiles
    const bool want_patch = pipe == 8 && AL == A_M4 && BM == 64 && BN == 64 && (BL == B_IM2COL || BL == B_IM2COL_TAPS) && ctx->split_mode < 4; // (split modes 4..6 are kernels of their own)
    if (pipe == 8 && !want_patch) pipe = 1;                            // the patch kernels: prepacked weights, 64x64 tiles, im2col B; anything else (and a launch they refuse) runs variant 3
    if constexpr (BL == B_IM2COL_TAPS) {
        if (pipe == 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "internal: tap-mask im2col needs an LDS-DMA pipeline");
    }

    const int nblk = (a.K + 255) / 256;
    const int T = a.tiles_m * a.tiles_n;
    int ntail = 0, t1 = T, S = 1;
    a.split_s = 1;
    a.order = ctx->tile_order & 3;
    const bool relaxed_split = (ctx->tile_order & 8) != 0 && (pipe == 1 || pipe == 3 || pipe == 4 || pipe == 7); // igemm_f32_dma_kernel only
    int split_mode = ctx->split_mode, split_req = ctx->split_s;
    if (pipe == 6 && ctx->wave_flavour >= 4) split_mode = 0; // (no split-K form: whole tiles only)
    if (split_mode == 3) { // auto: too few tiles to fill the chip -> cut every tile so that ~num_cus workgroups exist
        const long long wgs = (long long)T * Z;
        split_mode = (multi && wgs * 2 <= ctx->num_cus) ? 2 : 0;
        split_req = (int)(ctx->num_cus / (wgs > 0 ? wgs : 1));
    }
    if (multi && pipe != 2 && pipe != 5 && split_mode > 0 && split_mode < 4 && split_req > 1) { // (the wave-specialised and 16x16x4 kernels have no split-K form)
        const int s_req = split_req < nblk ? split_req : nblk;
        const int G = (nblk + s_req - 1) / s_req;
        S = (nblk + G - 1) / G;
        t1 = split_mode == 2 ? 0 : (T / ctx->num_cus) * ctx->num_cus;
        if (S > 1 && t1 < T) {
            ntail = T - t1;
            a.split_t1 = t1; a.split_s = S; a.split_g = G; a.split_slots = relaxed_split ? S : nblk; a.split_ntail = ntail;
            if (relaxed_split) a.order |= 8;
            const size_t need = 4096 + (size_t)Z * ntail * nblk * BM * BN * sizeof(float);
            char *sc = (char *)rten_scratch(ctx, need);
            if (!sc) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "split-K slab allocation failed (or attempted during graph capture)");
            a.slab = (float *)(sc + 4096);
            // the last-arriving producer folds the tile in the same launch (split_finish); RTEN_HIP_DEBUG bit 0x80000 keeps the fixup kernel
            a.split_counters = (!(ctx->debug & 0x80000) && (long long)Z * ntail <= rten_hip_ctx::kSplitCounters) ? ctx->split_counters : nullptr;
        } else {
            t1 = T;
        }
    }

    auto launch = [&](int mode, unsigned gx, double fl, double by) -> int32_t {
        const dim3 grid(gx, (unsigned)Z);
        if constexpr (kDma && AL == A_M4 && BM == 64 && BN == 64) {
            if (pipe == 6) { // one wave per tile (gemm_f32_wave.hip)
                snprintf(kname, sizeof kname, "igemm_f32_wave_kernel<%d,%d,%d>", BL, mode, ctx->wave_flavour);
                ProfScope ps(ctx, kname, fl, by);
                return rten_launch_gemm_f32_wave(ctx, &a, gx, (unsigned)Z, BL, mode, ctx->wave_flavour);
            }
            if constexpr (BL == B_IM2COL || BL == B_IM2COL_TAPS) {
                if (pipe == 8) { // image patches instead of per-element gathers (gemm_f32_patch.hip): 3x3 / stride 1 / padding 1 only
                    snprintf(kname, sizeof kname, "igemm_f32_patch_kernel<%d>", mode);
                    ProfScope ps(ctx, kname, fl, by);
                    const int32_t rc = rten_launch_gemm_f32_patch(ctx, &a, gx, (unsigned)Z, mode);
                    if (rc != RTEN_HIP_ERR_UNSUPPORTED) return rc;
                    pipe = 1; // a geometry the patch family does not cover: the three-stage LDS-DMA kernel
                }
            }
        }
        TRACE_ASSIGN(a, gx * (unsigned)Z);
        if constexpr (kDma) {
            if (pipe == 2) {
                snprintf(kname, sizeof kname, "igemm_f32_ws_kernel<%d,%d,%d,%s>", BM, BN, BL, mode == 1 ? "true" : "false");
                ProfScope ps(ctx, kname, fl, by);
                if (mode == 1) hipLaunchKernelGGL((igemm_f32_ws_kernel<BM, BN, BL, true>), grid, dim3(2 * NTHREADS), 0, ctx->stream, a);
                else hipLaunchKernelGGL((igemm_f32_ws_kernel<BM, BN, BL, false>), grid, dim3(2 * NTHREADS), 0, ctx->stream, a);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_ws_kernel launch");
                return RTEN_HIP_OK;
            }
            if (pipe == 1) {
                snprintf(kname, sizeof kname, "igemm_f32_dma_kernel<%d,%d,%d,%d,%d,3>", BM, BN, AL, BL, mode);
                ProfScope ps(ctx, kname, fl, by);
                const size_t pad = occupancy_pad(ctx, 3 * BK * (BM + BN) * 4);
                if (mode == 2) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 2>), grid, dim3(NTHREADS), pad, ctx->stream, a);
                else if (mode == 1) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 1>), grid, dim3(NTHREADS), pad, ctx->stream, a);
                else hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 0>), grid, dim3(NTHREADS), pad, ctx->stream, a);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma_kernel launch");
                return RTEN_HIP_OK;
            }
            if (pipe == 4) { // fragments first, MFMAs back to back (see MFK)
                snprintf(kname, sizeof kname, "igemm_f32_dma_kernel<%d,%d,%d,%d,%d,3,1>", BM, BN, AL, BL, mode);
                ProfScope ps(ctx, kname, fl, by);
                if (mode == 2) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 2, 3, 1>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                else if (mode == 1) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 1, 3, 1>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                else hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 0, 3, 1>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma_kernel launch");
                return RTEN_HIP_OK;
            }
            if (pipe == 5) { // 16x16x4 MFMAs: four independent accumulators per 32x32 of a wave's share
                snprintf(kname, sizeof kname, "igemm_f32_dma16_kernel<%d,%d,%d,%d,%d>", BM, BN, AL, BL, mode);
                ProfScope ps(ctx, kname, fl, by);
                if (mode == 1) hipLaunchKernelGGL((igemm_f32_dma16_kernel<BM, BN, AL, BL, 1>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                else hipLaunchKernelGGL((igemm_f32_dma16_kernel<BM, BN, AL, BL, 0>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma16_kernel launch");
                return RTEN_HIP_OK;
            }
            if constexpr (BM == 64 && BN == 64) {
                if (pipe == 7) { // TWO LDS stages (16 KB per workgroup): up to 7 workgroups per compute unit instead of 6 -- the other workgroups are the prefetch depth
                    snprintf(kname, sizeof kname, "igemm_f32_dma_kernel<%d,%d,%d,%d,%d,2>", BM, BN, AL, BL, mode);
                    ProfScope ps(ctx, kname, fl, by);
                    if (mode == 2) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 2, 2>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                    else if (mode == 1) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 1, 2>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                    else hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 0, 2>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                    RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma_kernel launch");
                    return RTEN_HIP_OK;
                }
            }
            if (pipe == 3) { // four LDS stages: three k-tiles in flight behind the one being multiplied
                snprintf(kname, sizeof kname, "igemm_f32_dma_kernel<%d,%d,%d,%d,%d,4>", BM, BN, AL, BL, mode);
                ProfScope ps(ctx, kname, fl, by);
                if (mode == 2) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 2, 4>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                else if (mode == 1) hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 1, 4>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                else hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 0, 4>), grid, dim3(NTHREADS), 0, ctx->stream, a);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma_kernel launch");
                return RTEN_HIP_OK;
            }
        }
        if constexpr (BL != B_IM2COL_TAPS) {
            snprintf(kname, sizeof kname, "igemm_f32_kernel<%d,%d,%d,%d,%d>", BM, BN, AL, BL, mode);
            ProfScope ps(ctx, kname, fl, by);
            if (mode == 2) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, AL, BL, 2>), grid, dim3(NTHREADS), 0, ctx->stream, a);
            else if (mode == 1) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, AL, BL, 1>), grid, dim3(NTHREADS), 0, ctx->stream, a);
            else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, AL, BL, 0>), grid, dim3(NTHREADS), 0, ctx->stream, a);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_kernel launch");
        }
        return RTEN_HIP_OK;
    };

    // Lean persistent plan (split mode 6, groups = resident workgroups per compute unit): igemm_f32_lean_kernel, 64x64 tiles, for
    // the convolution form (prepacked weights, one group, alpha = 1 / beta = 0) with K a multiple of 32; other calls ignore it.
    if constexpr (AL == A_M4 && (BL == B_N4 || BL == B_IM2COL_TAPS) && BM == 64 && BN == 64) {
        if (ctx->split_mode == 6 && Z == 1 && a.batch_inner <= 1 && a.alpha == 1.f && a.beta == 0.f && a.bias_kind != RTEN_HIP_BIAS_PER_COL &&
            a.K % LBK == 0 && a.K >= LBK && T > 1 && a.a_bs == (long long)a.K * a.a_cs) {
            // groups = workgroups per compute unit + 10 * (LDS stages - 3): 1..3 (three stages), 11, 12 (four), 21, 22 (five)
            const int nst = 3 + (ctx->split_s / 10 > 2 ? 2 : ctx->split_s / 10);
            int per_cu = ctx->split_s % 10;
            const int max_cu = 160 * 1024 / (nst * LBK * 128 * 4);
            per_cu = per_cu < 1 ? 1 : (per_cu > max_cu ? max_cu : per_cu);
            long long G = (long long)ctx->num_cus * per_cu;
            if (G > T) G = T;
            a.split_slots = ctx->num_cus; // (unused by this kernel's arithmetic: carries num_cus for the de-phasing delay)
            a.debug = (ctx->tile_order & 2) ? 0x100 : 0; // order bit 1 = de-phase co-resident workgroups (tuning knob)
            snprintf(kname, sizeof kname, "igemm_f32_lean_kernel<%d,%d>", BL, nst);
            ProfScope ps(ctx, kname, flops, bytes);
            auto go = [&](auto kern, int kStatic) {
                int dyn = (160 * 1024 / per_cu - kStatic - 256) & ~1023; // exactly per_cu workgroups fit a compute unit (see mode 5)
                if (dyn < 0) dyn = 0;
                if (kStatic + dyn > 64 * 1024 || kStatic > 48 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
                hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NTHREADS), (size_t)dyn, ctx->stream, a);
            };
            if (nst == 5) go(igemm_f32_lean_kernel<BL, 5>, 5 * LBK * 128 * 4);
            else if (nst == 4) go(igemm_f32_lean_kernel<BL, 4>, 4 * LBK * 128 * 4);
            else go(igemm_f32_lean_kernel<BL, 3>, 3 * LBK * 128 * 4);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_lean_kernel launch");
            return RTEN_HIP_OK;
        }
    }

    // Persistent plan (split mode 5, groups = resident workgroups per compute unit): one launch of num_cus x groups workgroups
    // that walk the tile list with the tile DMA running across tile boundaries (igemm_f32_pers_kernel).
    if constexpr (kDma) {
        if (ctx->split_mode == 5 && (pipe == 1 || pipe == 4 || pipe == 5) && T > 1) {
            int per_cu = ctx->split_s < 1 ? 1 : (ctx->split_s > 4 ? 4 : ctx->split_s);
            long long G = (long long)ctx->num_cus * per_cu;
            if (G > T) G = T;
            snprintf(kname, sizeof kname, "igemm_f32_pers_kernel<%d,%d,%d,%d,%s,%d>", BM, BN, AL, BL, pipe == 5 ? "true" : "false", pipe == 4 ? 1 : 0);
            ProfScope ps(ctx, kname, flops, bytes);
            const dim3 grid((unsigned)G, (unsigned)Z);
            // The dispatcher places workgroups wherever a slot is free: a grid of num_cus x R workgroups only lands R per compute
            // unit if no compute unit can take more.  Pad the LDS request (dynamic bytes the kernel never touches) so that exactly
            // `per_cu` workgroups fit into a compute unit's 160 KiB.
            constexpr int kStatic = 3 * BK * (BM + BN) * 4;
            int dyn = (160 * 1024 / per_cu - kStatic - 256) & ~1023;
            if (dyn < 0) dyn = 0;
            auto go = [&](auto kern) {
                if (kStatic + dyn > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
                hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), (size_t)dyn, ctx->stream, a);
            };
            if (pipe == 5) go(igemm_f32_pers_kernel<BM, BN, AL, BL, true>);
            else if (pipe == 4) go(igemm_f32_pers_kernel<BM, BN, AL, BL, false, 1>);
            else go(igemm_f32_pers_kernel<BM, BN, AL, BL, false>);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_pers_kernel launch");
            return RTEN_HIP_OK;
        }
    }

    // Thin-tile tail plan (split mode 4; convolution form alpha = 1, beta = 0, one group): the whole rounds of num_cus tiles go
    // to this tile shape over columns [0, n_big), the remaining columns to the 16x64 kernel on 16x16x4 MFMAs, whose quarter-size
    // per-SIMD blocks finish the tail in a quarter of a round (see igemm_f32_thin_kernel).  Same bits as any other plan.
    if constexpr (kDma && AL == A_M4) {
        if (ctx->split_mode == 4 && Z == 1 && a.batch_inner <= 1 && a.alpha == 1.f && a.beta == 0.f && a.bias_kind != RTEN_HIP_BIAS_PER_COL) {
            const long long rounds = T / ctx->num_cus;
            const long long big_cols = rounds * ctx->num_cus / a.tiles_m; // column tiles that fit into the whole rounds
            const long long big_tiles = big_cols * a.tiles_m;
            if (rounds >= 1 && big_cols >= 1 && big_cols * BN < a.N) {
                const int n_big = (int)(big_cols * BN);
                GemmArgs th = a;
                th.n_lo = n_big;
                th.tiles_m = (a.M + 15) / 16;
                const int thin_tiles_n = (a.N - n_big + 63) / 64;
                a.N = n_big; // the whole tiles stop at n_big; output addressing is unchanged
                a.tiles_n = n_big / BN;
                const double frac_big = (double)n_big / th.N;
                const int32_t rc = launch(multi ? 1 : 0, (unsigned)big_tiles, flops * frac_big, bytes * frac_big);
                if (rc) return rc;
                snprintf(kname, sizeof kname, "igemm_f32_thin_kernel<%d,%s>", BL, multi ? "true" : "false");
                ProfScope ps(ctx, kname, flops * (1.0 - frac_big), bytes * (1.0 - frac_big));
                const dim3 grid((unsigned)(th.tiles_m * thin_tiles_n), 1u);
                if (multi) hipLaunchKernelGGL((igemm_f32_thin_kernel<BL, true>), grid, dim3(NTHREADS), 0, ctx->stream, th);
                else hipLaunchKernelGGL((igemm_f32_thin_kernel<BL, false>), grid, dim3(NTHREADS), 0, ctx->stream, th);
                RTEN_LAUNCH_CHECK(ctx, "igemm_f32_thin_kernel launch");
                return RTEN_HIP_OK;
            }
        }
    }

    bool mixed = false;
    if constexpr (kDma && BM * BN < 128 * 128) mixed = pipe == 1 && ntail > 0 && t1 > 0;
    if (mixed) { // whole tiles and the tail's split-K producers in ONE launch (co-resident), then the fixup
        if constexpr (kDma && BM * BN < 128 * 128) {
            snprintf(kname, sizeof kname, "igemm_f32_dma_kernel<%d,%d,%d,%d,3,3>", BM, BN, AL, BL);
            ProfScope ps(ctx, kname, flops, bytes);
            TRACE_ASSIGN(a, (unsigned)(t1 + ntail * S) * (unsigned)Z);
            hipLaunchKernelGGL((igemm_f32_dma_kernel<BM, BN, AL, BL, 3>), dim3((unsigned)(t1 + ntail * S), (unsigned)Z), dim3(NTHREADS),
                               occupancy_pad(ctx, 3 * BK * (BM + BN) * 4), ctx->stream, a);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_dma_kernel (mixed) launch");
        }
    } else if (t1 > 0) {
        const int32_t rc = launch(multi ? 1 : 0, (unsigned)t1, flops * t1 / T, bytes * t1 / T);
        if (rc) return rc;
    }
    if (ntail > 0) {
        if (!mixed) {
            const int32_t rc = launch(2, (unsigned)(ntail * S), flops * ntail / T, bytes * ntail / T);
            if (rc) return rc;
        }
        if (!a.split_counters) {
            snprintf(kname, sizeof kname, "igemm_f32_fixup_kernel<%d,%d>", BM, BN);
            ProfScope ps(ctx, kname, 0.0, 8.0 * Z * ntail * nblk * BM * BN);
            hipLaunchKernelGGL((igemm_f32_fixup_kernel<BM, BN>), dim3((unsigned)ntail * 4u, (unsigned)Z), dim3(64), 0, ctx->stream, a);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_fixup_kernel launch");
        }
    }
    return RTEN_HIP_OK;
}

template <int AL, int BL>
int32_t launch_variant(rten_hip_ctx *ctx, GemmArgs &a, int Z, int cfg) {
    switch (cfg) {
    case 0: return launch_cfg<128, 128, AL, BL>(ctx, a, Z);
    case 1: return launch_cfg<128, 64, AL, BL>(ctx, a, Z);
    case 2: return launch_cfg<64, 128, AL, BL>(ctx, a, Z);
    default: return launch_cfg<64, 64, AL, BL>(ctx, a, Z);
    }
}

int pick_cfg(rten_hip_ctx *ctx, int M, long long N, int Z) {
    if (ctx->gemm_variant_override >= 24 && ctx->gemm_variant_override <= 32) return 3; // small-M streaming (31: calls that are not its shape), wave-tile kernels (24..26, 28..29), the two-stage ring (27), image patches (30): 64x64 plans
    if (ctx->gemm_variant_override >= 0 && ctx->gemm_variant_override < 24) return ctx->gemm_variant_override & 3;
    int best = 3;
    double best_cost = 1e300;
    for (int c = 0; c < 4; c++) {
        const TileCfg &t = kCfgs[c];
        const long long tiles = (long long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn) * Z;
        const long long rounds = (tiles + ctx->num_cus - 1) / ctx->num_cus;
        const double cost = (double)rounds * t.bm * t.bn * t.penalty;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

int32_t dispatch(rten_hip_ctx *ctx, GemmArgs &a, int Z, int al, int bl) {
    const int cfg = pick_cfg(ctx, a.M, a.N, Z);
    if (al == A_M4 && bl == B_N4) return launch_variant<A_M4, B_N4>(ctx, a, Z, cfg);
    if (al == A_M4 && bl == B_IM2COL) return launch_variant<A_M4, B_IM2COL>(ctx, a, Z, cfg);
    if (al == A_M4 && bl == B_IM2COL_TAPS) return launch_variant<A_M4, B_IM2COL_TAPS>(ctx, a, Z, cfg);
    if (al == A_K4 && bl == B_N4) return launch_variant<A_K4, B_N4>(ctx, a, Z, cfg);
    if (al == A_K4 && bl == B_K4) return launch_variant<A_K4, B_K4>(ctx, a, Z, cfg);
    if (bl == B_IM2COL) return launch_variant<A_SCALAR, B_IM2COL>(ctx, a, Z, cfg);
    return launch_variant<A_SCALAR, B_SCALAR>(ctx, a, Z, cfg);
}

// Largest byte offset (exclusive) an operand slice with the given extents/strides can touch.
long long extent_bytes(long long rows, long long rs, long long cols, long long cs) {
    if (rows <= 0 || cols <= 0) return 4;
    return ((rows - 1) * rs + (cols - 1) * cs + 1) * 4;
}

constexpr long long kMaxBufBytes = 0x7fffffffll; // buffer offsets are 32-bit; the OOB marker is 2^31

} // namespace

// Variants 0..3: tile shapes {128x128, 128x64, 64x128, 64x64} with the LDS-DMA pipeline on the conv paths;
// variants 4..7: the same tile shapes with the register-staged pipeline; variants 8..11: LDS-DMA with
// wave specialisation (4 MFMA waves + 4 loader waves); variants 12..15: LDS-DMA with four LDS stages.  Non-conv
// operand layouts always use the register-staged kernel.
// Variants 16..19: LDS-DMA, fragments-first MFMA issue; variants 20..23: LDS-DMA on 16x16x4 MFMAs.
// Variants 24..26: one wave per 64x64 tile (gemm_f32_wave.hip), k-tiles x LDS stages = 16 x 2, 8 x 4, 16 x 3; 27: 64x64 LDS-DMA with TWO stages;
// 28..29: one wave per 32x32 tile (barrier-free form of the 64x64 / 4-wave granularity; dense B), 16 x 2 and 16 x 3;
// 30: 3x3 / stride 1 / padding 1 convolutions with B staged as image patches (gemm_f32_patch.hip); every other launch runs as variant 3.
// 31: small-M weight streaming (rten_hip_gemm_f32 with one batch and M <= 64: gemm_f32_smallm_kernel; also what -1 = automatic picks there); every other launch runs as variant 3.
// 32: 3-channel 7x7 stride-2 convolutions with prepacked weights as a direct implicit GEMM (gemm_f32_stem.hip: the input patch of a 16 x 16 output tile in LDS); every other launch runs as variant 3.
RTEN_EXPORT int32_t rten_hip_num_gemm_variants(void) { return 33; }

RTEN_EXPORT int32_t rten_hip_set_gemm_variant_override(rten_hip_ctx *ctx, int32_t variant) {
    RTEN_CHECK_CTX(ctx);
    ctx->gemm_variant_override = variant;
    ctx->wave_flavour = (variant >= 24 && variant < 27) ? variant - 24 : (variant >= 28 && variant < 30) ? variant - 24 : 0;
    ctx->pipeline = variant == 30 ? 8 : variant == 27 ? 7 : (variant >= 24 && variant < 30) ? 6 : (variant >= 20 && variant < 24) ? 5 : (variant >= 16 && variant < 20) ? 4 : (variant >= 12 && variant < 16) ? 3 : (variant >= 8 && variant < 12) ? 2 : ((variant >= 4 && variant < 8) ? 0 : 1);
    return RTEN_HIP_OK;
}

// Exact split-K plan: mode 0 = off, 1 = split only the tiles past the last full round of num_cus workgroups (tail
// balancing), 2 = split every tile, 3 = automatic (default: split every tile when fewer than num_cus/2 workgroups
// would exist); `groups` = K groups per split tile (modes 1, 2).  Mode 4 is not a K split: whole rounds of tiles plus a tail of
// thin 16x64 tiles on 16x16x4 MFMAs (convolutions; other calls run their plain plan).
RTEN_EXPORT int32_t rten_hip_set_gemm_split(rten_hip_ctx *ctx, int32_t mode, int32_t groups) {
    RTEN_CHECK_CTX(ctx);
    if (mode < 0 || mode > 6 || groups < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "set_gemm_split: bad mode/groups");
    ctx->split_mode = mode;
    ctx->split_s = groups;
    return RTEN_HIP_OK;
}

// Workgroup -> tile order (tuning knob, sticky): bit 0 = tiles walk n fastest instead of m fastest; bit 1 = split-K
// workgroups walk tiles fastest and K groups slowest (each XCD's L2 then holds one K slice of both operands).
RTEN_EXPORT int32_t rten_hip_set_gemm_order(rten_hip_ctx *ctx, int32_t order) {
    RTEN_CHECK_CTX(ctx);
#ifdef RTEN_ABLATION // bit 3 = relaxed split-K (one partial per K group: NOT the reference's order, NOT bit-exact): measurement builds only
    constexpr int allowed = 0x7b;
#else
    constexpr int allowed = 0x73;
#endif
    if (order < 0 || (order & ~allowed)) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "set_gemm_order: bits 0-1 (tile walk) and 4-6 (workgroups per compute unit) only (bit 3, the relaxed split-K, exists in -DRTEN_ABLATION measurement builds)");
    ctx->tile_order = order;
    return RTEN_HIP_OK;
}

namespace {
int32_t gemm_f32_entry(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c, bool allow_gemv);
}

RTEN_EXPORT int32_t rten_hip_gemm_f32(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b,
                                      const float *bias, float *c) {
    RTEN_CHECK_CTX(ctx);
    return gemm_f32_entry(ctx, d, a, b, bias, c, ctx->gemv_order != 0);
}

int32_t rten_gemm_f32_blocked(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c) {
    RTEN_CHECK_CTX(ctx);
    return gemm_f32_entry(ctx, d, a, b, bias, c, false);
}

RTEN_EXPORT int32_t rten_hip_set_gemv_order(rten_hip_ctx *ctx, int32_t on, int32_t reference_threads) {
    RTEN_CHECK_CTX(ctx);
    if (reference_threads < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    ctx->gemv_order = on ? 1 : 0;
    ctx->gemv_threads = reference_threads;
    return RTEN_HIP_OK;
}

namespace {
// Small-M weight streaming (gemm_f32_smallm_kernel): one batch, M <= 64.  Returns RTEN_HIP_ERR_UNSUPPORTED when the call is not its shape.
int32_t launch_smallm(rten_hip_ctx *ctx, GemmArgs &a, const rten_hip_gemm_desc *d, const float *ap, const float *bp) {
    if (d->batch != 1 || d->m > 64 || d->k <= 0) return RTEN_HIP_ERR_UNSUPPORTED;
    const int MT = d->m <= 16 ? 1 : d->m <= 32 ? 2 : 4, RB = 16 * (4 / MT);
    const long long gx = (d->n + RB - 1) / RB, nblk = (d->k + 255) / 256;
    if (gx > 0x7fffffff || nblk > 65535) return RTEN_HIP_ERR_UNSUPPORTED;
    if (nblk > 1) {
        if (!ctx->split_counters || gx * 4 > rten_hip_ctx::kSplitCounters) return RTEN_HIP_ERR_UNSUPPORTED;
        char *sc = (char *)rten_scratch(ctx, 4096 + (size_t)gx * (size_t)nblk * 4096);
        // (growing the slab is impossible while a capture is active: decline, the tiled kernels run the product -- ADVICE round 5)
        if (!sc) return ctx->capturing ? RTEN_HIP_ERR_UNSUPPORTED : rten_set_error(ctx, RTEN_HIP_ERR_HIP, "split-K slab allocation failed");
        a.slab = (float *)(sc + 4096);
        a.split_counters = ctx->split_counters;
    }
    const bool k4 = d->k % 4 == 0;
    // loader forms (16-byte loads): bit 0 / 1 = A / B depth-contiguous; bit 2 / 3 = A rows / B columns contiguous
    a.tiles_m = ((d->a_cs == 1 && d->a_rs % 4 == 0 && k4 && aligned16(ap)) ? 1 : 0) | ((d->b_rs == 1 && d->b_cs % 4 == 0 && k4 && aligned16(bp)) ? 2 : 0);
    if (!(a.tiles_m & 1) && d->a_rs == 1 && d->a_cs % 4 == 0 && d->m % 4 == 0 && aligned16(ap)) a.tiles_m |= 4;
    if (!(a.tiles_m & 2) && d->b_cs == 1 && d->b_rs % 4 == 0 && d->n % 4 == 0 && aligned16(bp)) a.tiles_m |= 8;
    char kname[64];
    snprintf(kname, sizeof kname, "gemm_f32_smallm_kernel<%d>", MT);
    ProfScope ps(ctx, kname, 2.0 * d->m * (double)d->n * d->k, 4.0 * ((double)d->m * d->k + (double)d->k * d->n + (double)d->m * d->n));
    const dim3 grid((unsigned)gx, (unsigned)nblk);
    if (MT == 1) hipLaunchKernelGGL((gemm_f32_smallm_kernel<1>), grid, dim3(256), 0, ctx->stream, a);
    else if (MT == 2) hipLaunchKernelGGL((gemm_f32_smallm_kernel<2>), grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((gemm_f32_smallm_kernel<4>), grid, dim3(256), 0, ctx->stream, a);
    RTEN_LAUNCH_CHECK(ctx, "gemm_f32_smallm_kernel launch");
    return RTEN_HIP_OK;
}

int32_t gemm_f32_entry(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c, bool allow_gemv) {
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm: negative dimension");
    if (d->bias_kind != RTEN_HIP_BIAS_NONE && !bias)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm: bias_kind set but bias is NULL");
    if (d->m == 0 || d->n == 0 || d->batch == 0) return RTEN_HIP_OK; // rten-gemm/src/lib.rs:835-839
    if (!c || (d->k > 0 && (!a || !b))) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm: NULL operand");
    if (d->ldc < d->n) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm: ldc < n");
    if (d->a_rs < 0 || d->a_cs < 0 || d->b_rs < 0 || d->b_cs < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "gemm: negative strides are not supported");
    // one row, B not prepacked: the reference takes its gemv kernels, whose accumulation order is not the blocked one (lib.rs:876-891)
    if (allow_gemv && d->m == 1 && d->k > 0) return rten_gemv_f32(ctx, d, a, b, bias, c);

    GemmArgs g = {};
    g.A = a ? a : c; g.B = b ? b : c; g.C = c; g.bias = bias; g.res = nullptr;
    g.M = d->m; g.N = d->n; g.K = d->k;
    g.a_rs = d->a_rs; g.a_cs = d->a_cs; g.a_bs = d->a_bs;
    g.b_rs = d->b_rs; g.b_cs = d->b_cs; g.b_ns = 0; g.b_bs = d->b_bs;
    g.c_rs = d->ldc; g.c_ns = 0; g.c_bs = d->c_bs;
    g.bias_bs = 0;
    g.batch_inner = d->batch_inner; g.a_bsi = d->a_bsi; g.b_bsi = d->b_bsi; g.c_bsi = d->c_bsi;
    g.Pn = d->n;
    g.alpha = d->alpha; g.beta = d->beta;
    g.bias_kind = d->bias_kind; g.act = d->act;
    g.a_dir_m = (d->a_rs == 1 && d->a_cs != 1) ? 1 : 0;
    g.b_dir_n = (d->b_cs == 1 || d->b_rs != 1) ? 1 : 0;
    const long long ab = extent_bytes(d->m, d->a_rs, d->k, d->a_cs), bb = extent_bytes(d->k, d->b_rs, d->n, d->b_cs);
    if (ab > kMaxBufBytes || bb > kMaxBufBytes || extent_bytes(d->m, d->ldc, d->n, 1) > kMaxBufBytes)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "gemm: operand slices above 2 GiB are not supported");
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;

    // few rows: stream B through every compute unit instead of a handful of 64-row tiles (variant 31; the automatic choice; RTEN_HIP_DEBUG bit 0x40000 = off)
    if ((ctx->gemm_variant_override == 31 || (ctx->gemm_variant_override < 0 && !(ctx->debug & 0x40000))) && d->m <= 64 && d->batch == 1 && d->k > 0) {
        const int32_t rc = launch_smallm(ctx, g, d, a, b);
        if (rc != RTEN_HIP_ERR_UNSUPPORTED) return rc;
    }

    int al = A_SCALAR, bl = B_SCALAR;
    if (d->k > 0) {
        const bool a4 = d->a_bs % 4 == 0 && d->a_bsi % 4 == 0 && aligned16(a), b4 = d->b_bs % 4 == 0 && d->b_bsi % 4 == 0 && aligned16(b);
        if (d->a_rs == 1 && d->a_cs % 4 == 0 && d->m % 4 == 0 && a4) al = A_M4;
        else if (d->a_cs == 1 && d->a_rs % 4 == 0 && d->k % 4 == 0 && a4) al = A_K4;
        if (d->b_cs == 1 && d->b_rs % 4 == 0 && d->n % 4 == 0 && b4) bl = B_N4;
        else if (d->b_rs == 1 && d->b_cs % 4 == 0 && d->k % 4 == 0 && b4) bl = B_K4;
        const bool have = (al == A_M4 && bl == B_N4) || (al == A_K4 && bl == B_N4) || (al == A_K4 && bl == B_K4);
        if (!have) { al = A_SCALAR; bl = B_SCALAR; }
    }
    return dispatch(ctx, g, d->batch, al, bl);
}
} // namespace

// ---- conv weight staging: W[g][m][k] (OIHW) -> packed[g][k][Og4], zero padded (Og4 = round_up(O/g, 4))
namespace {
__global__ void conv_prepack_f32_kernel(const float *__restrict__ w, float *__restrict__ packed, int groups, int Og,
                                        int K, int Og4) {
    const long long total = (long long)groups * K * Og4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i % Og4);
        const long long r = i / Og4;
        const int k = (int)(r % K);
        const int g = (int)(r / K);
        packed[i] = m < Og ? w[((long long)g * Og + m) * K + k] : 0.f;
    }
}

int32_t check_conv_desc(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d) {
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->groups <= 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Group count must be > 0");
    if (d->c % d->groups != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Input channel count not divisible by groups");
    if (d->o % d->groups != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Output channel count not divisible by groups");
    if (d->n < 0 || d->c <= 0 || d->h <= 0 || d->w <= 0 || d->o <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 ||
        d->stride_w <= 0 || d->dil_h <= 0 || d->dil_w <= 0 || d->out_h < 0 || d->out_w < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv: invalid geometry");
    const long long in_elems = (long long)d->n * d->c * d->h * d->w;
    const long long out_elems = (long long)d->n * d->o * d->out_h * d->out_w;
    if (in_elems * 4 > kMaxBufBytes || out_elems * 4 > kMaxBufBytes)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv: tensors above 2 GiB are not supported");
    if ((long long)d->kh * d->dil_h >= 0x7fff || (long long)d->kw * d->dil_w >= 0x7fff || d->h >= 0x7fff || d->w >= 0x7fff)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv: spatial extent above 32766 is not supported");
    return RTEN_HIP_OK;
}

// im2col LUT cache (per context): one table per (Cg, kh, kw, dil, H, W)
const i32x2 *get_im2col_lut(rten_hip_ctx *ctx, int Cg, int kh, int kw, int dy, int dx, int H, int W, int taps) {
    char key[96];
    snprintf(key, sizeof key, "%d.%d.%d.%d.%d.%d.%d.%d", Cg, kh, kw, dy, dx, H, W, taps);
    auto it = ctx->luts.find(key);
    if (it != ctx->luts.end()) return (const i32x2 *)it->second;
    if (ctx->capturing) return nullptr; // allocation is not capturable: warm up eagerly first
    const int K = Cg * kh * kw;
    const int Kpad = ((K + BK - 1) / BK + MAX_NSTAGE + 2) * BK; // tile / LUT look-ahead runs up to NSTAGE tiles past the end
    void *dptr = nullptr;
    if (hipMalloc(&dptr, (size_t)Kpad * sizeof(i32x2)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(im2col_lut_kernel, dim3((Kpad + 255) / 256), dim3(256), 0, ctx->stream, (i32x2 *)dptr, K, Kpad, kh * kw,
                       kw, H * W, W, dy, dx, taps);
    ctx->luts[key] = dptr;
    return (const i32x2 *)dptr;
}
} // namespace

// gemm_f32_stem.hip (GEMM variant 32)
bool rten_small_c_conv_f32_supported(const rten_hip_conv2d_desc *d, int weights_packed, const float *residual);
int32_t rten_small_c_conv_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w_packed, const float *bias, uint32_t flags, float *y);

// depthwise.hip
int32_t rten_depthwise_conv2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w, int32_t weights_packed, const float *bias,
                                  const float *residual, uint32_t flags, float *y);

RTEN_EXPORT size_t rten_hip_conv2d_f32_packed_bytes(const rten_hip_conv2d_desc *d) {
    if (!d || d->groups <= 0) return 0;
    const int Og = d->o / d->groups, Og4 = (Og + 3) & ~3;
    const long long K = (long long)(d->c / d->groups) * d->kh * d->kw;
    return (size_t)d->groups * K * Og4 * sizeof(float);
}

RTEN_EXPORT int32_t rten_hip_conv2d_f32_prepack(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *w,
                                                float *packed) {
    RTEN_CHECK_CTX(ctx);
    int32_t rc = check_conv_desc(ctx, d);
    if (rc) return rc;
    if (!w || !packed) return RTEN_HIP_ERR_INVALID_VALUE;
    const int Og = d->o / d->groups, Og4 = (Og + 3) & ~3;
    const int K = (d->c / d->groups) * d->kh * d->kw;
    const long long total = (long long)d->groups * K * Og4;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_prepack_f32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, w, packed, d->groups, Og, K,
                       Og4);
    RTEN_LAUNCH_CHECK(ctx, "conv_prepack_f32_kernel launch");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_conv2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w,
                                        int32_t weights_packed, const float *bias, const float *residual,
                                        uint32_t flags, float *y) {
    RTEN_CHECK_CTX(ctx);
    int32_t rc = check_conv_desc(ctx, d);
    if (rc) return rc;
    if (!x || !w || !y) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv: NULL operand");
    if ((flags & RTEN_HIP_CONV_RESIDUAL) && !residual)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv: residual flag without residual tensor");
    if (d->n == 0 || d->out_h == 0 || d->out_w == 0) return RTEN_HIP_OK;
    {   // dispatch order of conv_impl (conv.rs:248-284): the pointwise GEMM first (groups == 1 only), then depthwise
        const bool pw = d->kh == 1 && d->kw == 1 && d->groups == 1 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 &&
                        d->pads[0] == 0 && d->pads[1] == 0 && d->pads[2] == 0 && d->pads[3] == 0;
        if (!pw && d->c == d->o && d->groups == d->c) return rten_depthwise_conv2d_f32(ctx, d, x, w, weights_packed, bias, residual, flags, y);
    }
    // variant 32 (also what -1 = automatic picks there): 3-channel 7x7 stride-2 convolutions (a ResNet stem) as a direct implicit GEMM over an image patch in LDS
    // (gemm_f32_stem.hip); same bits
    if ((ctx->gemm_variant_override == 32 || ctx->gemm_variant_override < 0) && aligned16(w) && rten_small_c_conv_f32_supported(d, weights_packed, (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr))
        return rten_small_c_conv_f32(ctx, d, x, w, bias, flags, y);
    const int Cg = d->c / d->groups, Og = d->o / d->groups, Og4 = (Og + 3) & ~3;
    const int K = Cg * d->kh * d->kw;
    const int P = d->out_h * d->out_w;
    const long long HW = (long long)d->h * d->w;
    if ((long long)K * Og4 * 4 > kMaxBufBytes)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv: weights above 2 GiB per group are not supported");

    GemmArgs g = {};
    g.A = w; g.B = x; g.C = y; g.bias = bias; g.res = (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr;
    g.M = Og; g.N = d->n * P; g.K = K;
    int al;
    if (weights_packed) {
        g.a_rs = 1; g.a_cs = Og4; g.a_bs = (long long)K * Og4;
        al = aligned16(w) ? A_M4 : A_SCALAR;
        g.a_dir_m = 1;
        g.a_bytes = (unsigned)((long long)K * Og4 * 4);
    } else {
        g.a_rs = K; g.a_cs = 1; g.a_bs = (long long)Og * K;
        al = A_SCALAR;
        g.a_dir_m = 0;
        g.a_bytes = (unsigned)((long long)Og * K * 4);
    }
    g.b_bs = (long long)Cg * HW;
    g.b_ns = (long long)d->c * HW;
    // one group's slice spans from its first channel of image 0 to its last channel of the last image
    g.b_bytes = (unsigned)((((long long)(d->n - 1) * d->c + Cg) * HW) * 4);
    g.c_rs = P; g.c_ns = (long long)d->o * P; g.c_bs = (long long)Og * P;
    g.bias_bs = Og;
    g.Pn = P;
    g.alpha = 1.f; g.beta = 0.f;
    g.bias_kind = bias ? RTEN_HIP_BIAS_PER_ROW : RTEN_HIP_BIAS_NONE;
    g.act = (flags & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;

    const bool pointwise = d->kh == 1 && d->kw == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pads[0] == 0 &&
                           d->pads[1] == 0 && d->pads[2] == 0 && d->pads[3] == 0; // conv.rs:250-258 (dilation is moot)
    int bl = B_IM2COL;
    if (pointwise && al == A_M4 && (P % 4) == 0 && aligned16(x)) {
        // the image batch is a two-level [C, (n, H*W)] matrix: no gather needed
        bl = B_N4;
        g.b_rs = HW; g.b_cs = 1;
    } else {
        // <= 31 kernel taps on an LDS-DMA pipeline: per-lane padding bitmask instead of per-gather bounds tests
        const int taps = (al == A_M4 && ctx->pipeline != 0 && d->kh * d->kw <= 31) ? 1 : 0;
        if (taps) bl = B_IM2COL_TAPS;
        g.KH = d->kh; g.KW = d->kw; g.dy = d->dil_h; g.dx = d->dil_w;
        g.lut = get_im2col_lut(ctx, Cg, d->kh, d->kw, d->dil_h, d->dil_w, d->h, d->w, taps);
        if (!g.lut) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "conv: im2col table allocation failed (warm up before graph capture)");
        g.H = d->h; g.W = d->w; g.OW = d->out_w;
        g.sy = d->stride_h; g.sx = d->stride_w;
        g.pt = d->pads[0]; g.pl = d->pads[1];
    }
    g.b_dir_n = 1;
    g.debug = ctx->debug;
    return dispatch(ctx, g, d->groups, al, bl);
}

#ifdef RTEN_TRACE
RtenTraceHost g_trace_host;
// Trace builds only: hand the kernels a device buffer of `cap` 128-byte records (NULL: off) / read the number of record slots handed out to launches so far.
RTEN_EXPORT int32_t rten_hip_debug_trace_set(rten_hip_ctx *ctx, void *buf, uint32_t cap) {
    RTEN_CHECK_CTX(ctx);
    g_trace_host.buf = (unsigned long long *)buf;
    g_trace_host.cap = cap;
    return RTEN_HIP_OK;
}
RTEN_EXPORT int32_t rten_hip_debug_trace_count(rten_hip_ctx *ctx, uint32_t *n) {
    RTEN_CHECK_CTX(ctx);
    *n = g_trace_host.next;
    return RTEN_HIP_OK;
}
#endif
