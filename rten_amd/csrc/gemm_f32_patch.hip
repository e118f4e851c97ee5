// 3x3 convolutions (stride 1, padding 1, dilation 1) whose B operand is staged as IMAGE PATCHES: the "patch" family of the f32 implicit-GEMM
// kernels (gfx950).
//
// Replaces the same reference code as gemm_f32.hip (rten-gemm/src/lib.rs:794-1093, kernels/simd_generic.rs:285-414, the virtual im2col of
// rten-gemm/src/im2col.rs:56-212 and conv_impl, src/ops/conv.rs:124-365) for A = prepacked k-major weights [K][M], B = the im2col matrix of a
// 3x3 / stride 1 / padding 1 convolution.
//
// Why (round 4; profiles/r07/wave_tile_probe.txt): the im2col kernels fetch every (k, pixel) element of the B tile with its
// own lane of a `buffer_load_dword ... lds` gather -- 16 gather instructions per 16-k tile and workgroup -- and a compute unit retires one such
// instruction in ~40 cycles whatever it hits: 640 cycles of gathers against 512 cycles of MFMAs per k-tile bound the 3x3 layers (45 % of
// ResNet-50's FLOPs) at 72-80 % of the dense loop.  But the nine taps of a channel read the SAME pixels nine times, shifted.  Here a k-tile is
// two whole channels (18 k = 2 x 9 taps), and what goes to LDS per channel is the raw image patch under the tile's 64 output pixels, ONCE:
// the rows of a virtual zero-padded image stack ([image][H + 2 rows][PW = W + pad dwords]; padding rows / columns are out-of-range lanes of the
// DMA, i.e. zeros, never branches), 1-2 `dwordx4` (W % 4 == 0) or 2-4 `dword` instructions per channel instead of 9.  The shift happens at the
// LDS READ: lane (pixel n, k parity) reads patch[base(n) + ky * PW + kx] -- a per-lane address that is loop invariant (nine VGPRs), consecutive
// pixels are consecutive dwords, so the reads are conflict-free.
//
// Numerics: the k order is untouched (k = c * 9 + ky * 3 + kx ascending, one FMA chain per output element per depth block of 256, blocks folded
// with separate adds: rten-gemm/src/lib.rs:630-633,1008-1013); k-tiles of 18 do not divide 256, so a depth-block boundary falls INSIDE a tile
// (always between two k-pairs: 256 j is even) and such "edge" tiles run a checked copy of the loop body that folds / parks the accumulator at the
// pair where the reference starts a new block.  Split-K groups start at depth-block boundaries exactly as in the other families and use the same
// slab image and last-arrival fold (split_finish), so plans are interchangeable.  Bit-identical to the oracle; the variant sweeps in tests/ run
// this family next to the others.
#include "gemm_f32_common.h"

namespace {

struct PatchGeom {
    int PW;          // dwords per virtual row (>= W + 1; multiple of 4 on the dwordx4 path)
    int HP;          // virtual rows per image = H + 2 (row 0 and row H + 1 are padding)
    int chan_stride; // elements between channels of one image (H * W)
    int n_images;    // images in one batch slice of B (pixels >= N are clamped, rows beyond the last image are padding)
    RtenDiv dPW, dHP;
};

constexpr int PK = 18;                  // k-tile: two channels x nine taps
constexpr int A_TILE = PK * 64;         // floats of the A stage that carry data ...
constexpr int A_STAGE = 5 * 256;        // ... and its LDS footprint: five dwordx4 instructions (the fifth half empty)

// MODE 0: K <= 256 (one depth block); 1: several depth blocks folded in registers; 2: split-K producer (one K group of one split tile, raw
// accumulators parked in the slab, last arrival folds).  G: dwords per DMA lane (4: dwordx4, W % 4 == 0; 1: dword).  NP: DMA instructions per
// channel patch (a patch slot holds NP * 64 * G dwords).  Two LDS stages, one barrier per k-tile (the form the tuner prefers among the 4-wave
// pipelines: variant 27).
template <int MODE, int G, int NP>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_f32_patch_kernel(const GemmArgs p, const PatchGeom g) {
    TR_DECL
    TR_STAMP(0)
    constexpr bool MULTI_KC = MODE == 1, SPLIT = MODE == 2;
    constexpr int PATCH = NP * 64 * G;        // dwords per channel slot
    constexpr int STAGE = A_STAGE + 2 * PATCH; // floats per stage: A [18][64] (+ pad), then two channel patches
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int z = blockIdx.y;

    int tile, grp = -1;
    {
        const int id = blockIdx.x, nt = (int)gridDim.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7; // XCD-chunked: each XCD (private L2) walks a contiguous run of tiles sharing a B panel
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        if constexpr (SPLIT) {
            const int rr = tile;
            if (p.order & 2) { grp = rr / p.split_ntail; tile = p.split_t1 + rr - grp * p.split_ntail; }
            else { tile = p.split_t1 + rr / p.split_s; grp = rr - (rr / p.split_s) * p.split_s; }
        }
    }
    const int bm = (p.order & 1) ? tile / p.tiles_n : tile % p.tiles_m, bn = (p.order & 1) ? tile % p.tiles_n : tile / p.tiles_m;
    const int m0 = bm * 64, n0 = bn * 64;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);

    // ---- A: instruction `wave` of round 0 covers rows 4 wave .. 4 wave + 3 of the [18][64] tile; round 1 (rows 16, 17) is wave 0's
    unsigned a_voff0, a_voff1;
    {
        const int f = (wave * 64 + lane) * 4;
        const int k = f >> 6, m = m0 + (f & 63);
        a_voff0 = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
        const int k1 = 16 + (lane >> 4), m1 = m0 + (lane & 15) * 4;
        a_voff1 = (lane < 32 && m1 < (int)p.a_cs) ? (unsigned)(((long long)k1 * p.a_cs + m1) * 4) : OOB;
    }
    const unsigned a_kstep = (unsigned)(PK * p.a_cs * 4);

    // ---- B: the tile's window of the virtual padded image stack.  Virtual dword index of pixel (img, oy, ox), tap (0, 0):
    //      u = (img * HP + oy) * PW + ox - 1   (u = -1 for the very first pixel: everything is kept shifted by +G so that it stays >= 0)
    auto pix_u = [&](int n) {
        const int nn = n < p.N ? n : p.N - 1;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const int oy = np / p.OW, ox = np - oy * p.OW;
        return (nb * g.HP + oy) * g.PW + ox - 1 + G;
    };
    const int w0 = (pix_u(n0) / G) * G; // first virtual dword (shifted) of the window, on a DMA-lane boundary
    unsigned b_voff[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int u = w0 + (i * 64 + lane) * G - G; // unshifted virtual dword of this lane's first element (>= -G)
        const int uu = u < 0 ? 0 : u;
        const int v = rten_div(uu, g.dPW), col = uu - v * g.PW;
        const int img = rten_div(v, g.dHP), r = v - img * g.HP;
        const bool ok = u >= 0 && col < p.W && r >= 1 && r <= p.H && img < g.n_images;
        b_voff[i] = ok ? (unsigned)(((long long)img * p.b_ns + (long long)(r - 1) * p.W + col) * 4) : OOB;
    }
    const unsigned b_cstep = (unsigned)g.chan_stride * 4u;
    // fragment addresses: this lane's pixel, the nine k-pairs of a tile (k = 2 pp + half: channel slot e = k / 9, tap = k % 9)
    int boff[PK / 2];
    {
        const int lb = pix_u(n0 + wn0 + l31) - w0;
#pragma unroll
        for (int pp = 0; pp < PK / 2; pp++) {
            const int k = 2 * pp + half, e = k >= 9 ? 1 : 0, tap = k - 9 * e;
            const int ky = tap / 3, kx = tap - 3 * ky;
            boff[pp] = A_STAGE + e * PATCH + lb + ky * g.PW + kx;
        }
    }

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + A_STAGE;
        const unsigned a_soff = (unsigned)kt * a_kstep;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + wave * 256), 16, (int)a_voff0, (int)a_soff, 0, 0);
        if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + 1024), 16, (int)a_voff1, (int)a_soff, 0, 0);
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int i = 0; i < NP; i++)
                if (wave == ((e * NP + i + 1) & 3)) { // dealt over the waves, starting behind wave 0's extra A instruction
                    if constexpr (G == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + e * PATCH + i * 256), 16, (int)b_voff[i], (int)((unsigned)(2 * kt + e) * b_cstep), 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + e * PATCH + i * 64), 4, (int)b_voff[i], (int)((unsigned)(2 * kt + e) * b_cstep), 0, 0);
                }
    };

    f32x16 acc[1][1];
    [[maybe_unused]] f32x16 tot[1][1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;
    };
    zero_acc();
    const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;

    // raw accumulator image of this tile in the split-K slab: [wave][quad][lane] float4 (the 4-wave kernels' image: split_finish<64, 64, 1, 1>)
    [[maybe_unused]] auto store_raw = [&](int slot) {
        int loff = wave * 1024 + lane * 4;
        asm volatile("" : "+v"(loff));
        float *base = p.slab + (((long long)z * p.split_ntail + (tile - p.split_t1)) * p.split_slots + slot) * (long long)(64 * 64);
        if (p.split_counters) { // folded in this launch, possibly on another XCD: write through (see coherent_store4)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 64 * 64 * 4, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 o = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
                coherent_store4(rs, (unsigned)(loff + q * 256) * 4u, o);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
            *(f32x4 *)(base + loff + q * 256) = o;
        }
    };
    // the reference starts a new depth block at k = 256 j: fold (or park) the block that ends there
    [[maybe_unused]] auto block_boundary = [&](int k) {
        if constexpr (SPLIT) {
            store_raw((k >> 8) - 1);
            zero_acc();
        } else if constexpr (MULTI_KC) {
            int mbv = mb, nbv = nb0;
            asm volatile("" : "+v"(mbv), "+v"(nbv));
            if (k == 256) fold_first<1, 1>(p, z, acc, tot, mbv, nbv, c_zoff);
            else fold_next<1, 1>(p, acc, tot);
            zero_acc();
        }
    };

    auto compute_fast = [&](int stage) {
        const float *S = smem + stage * STAGE;
        const float *As = S + wm0 + l31 + half * 64;
        float af[2], bf[2];
        af[0] = As[0];
        bf[0] = S[boff[0]];
#pragma unroll
        for (int pp = 0; pp < PK / 2; pp++) {
            const int cur = pp & 1, nxt = cur ^ 1;
            if (pp + 1 < PK / 2) {
                af[nxt] = As[2 * (pp + 1) * 64];
                bf[nxt] = S[boff[pp + 1]];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc[0][0], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
    };
    // tiles that hold the start / end of this workgroup's K range or a depth-block boundary: the same pairs, each behind its (uniform) tests
    auto compute_edge = [&](int stage, int kbase, int k_begin, int k_end) {
        const float *S = smem + stage * STAGE;
        const float *As = S + wm0 + l31 + half * 64;
#pragma unroll
        for (int pp = 0; pp < PK / 2; pp++) {
            const int k = kbase + 2 * pp;
            if (k < k_begin || k >= k_end) continue;
            if ((k & 255) == 0 && k > k_begin) block_boundary(k);
            const float a = As[2 * pp * 64], b = S[boff[pp]];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0][0], 0, 0, 0);
        }
    };

    const int nblk = (MULTI_KC || SPLIT) ? (p.K + 255) >> 8 : 1;
    int blk0 = 0, blk1 = nblk;
    if constexpr (SPLIT) {
        blk0 = grp * p.split_g;
        blk1 = blk0 + p.split_g < nblk ? blk0 + p.split_g : nblk;
    }
    const int k_begin = blk0 << 8, k_end = (blk1 << 8) < p.K ? (blk1 << 8) : p.K;
    const int kt_first = k_begin / PK, kt_last = (k_end + PK - 1) / PK;
    issue_tile(kt_first, 0);
    int stage = 0;
    TR_STAMP(1)
    for (int kt = kt_first; kt < kt_last; kt++) {
        wait_vmcnt<0>(); // this wave's share of tile kt has landed ...
        __builtin_amdgcn_s_barrier(); // ... everybody's has, and everybody is done reading the other stage (tile kt - 1)
#ifdef RTEN_TRACE
        if (tr_trips == 0) TR_STAMP(2)
        tr_trips++;
#endif
        if (kt + 1 < kt_last) issue_tile(kt + 1, stage ^ 1);
        const int kbase = kt * PK;
        const bool interior = kbase >= k_begin && kbase + PK <= k_end && ((kbase + PK - 1) >> 8) == (kbase >> 8) && ((kbase & 255) != 0 || kbase == k_begin);
        if (interior) compute_fast(stage);
        else compute_edge(stage, kbase, k_begin, k_end);
        stage ^= 1;
    }
    TR_STAMP(3)
    [[maybe_unused]] constexpr unsigned TR_KID = 64u | (64u << 8) | (MODE << 16) | (B_IM2COL_TAPS << 20) | (4u << 24) | (2u << 28);

    if constexpr (SPLIT) {
        store_raw(blk1 - 1);
        if (p.split_counters) split_finish<64, 64, 1, 1>(p, z, tile, wave, lane, m0, n0, c_zoff, reinterpret_cast<int *>(smem));
        TR_STAMP(4)
        TR_WRITE(TR_KID, tile, grp)
        return;
    } else {
        if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
            fold_next<1, 1>(p, acc, tot);
            store_out<1, 1>(p, tot, mb, nb0, c_zoff);
        } else {
            fold_first<1, 1>(p, z, acc, acc, mb, nb0, c_zoff);
            store_out<1, 1>(p, acc, mb, nb0, c_zoff);
        }
        TR_STAMP(4)
        TR_WRITE(TR_KID, tile, grp)
    }
}

template <int G, int NP>
int32_t launch_patch(rten_hip_ctx *ctx, const GemmArgs &a, const PatchGeom &g, dim3 grid, int mode) {
    switch (mode) {
    case 2: hipLaunchKernelGGL((igemm_f32_patch_kernel<2, G, NP>), grid, dim3(NTHREADS), 0, ctx->stream, a, g); break;
    case 1: hipLaunchKernelGGL((igemm_f32_patch_kernel<1, G, NP>), grid, dim3(NTHREADS), 0, ctx->stream, a, g); break;
    default: hipLaunchKernelGGL((igemm_f32_patch_kernel<0, G, NP>), grid, dim3(NTHREADS), 0, ctx->stream, a, g); break;
    }
    RTEN_LAUNCH_CHECK(ctx, "igemm_f32_patch_kernel launch");
    return RTEN_HIP_OK;
}

} // namespace

// Does the patch family cover this launch?  3x3 taps, stride 1, padding 1 (top / left; the output is as large as the input), dilation 1, K a whole
// number of channel pairs, and every tile's window of the virtual image stack fits a patch slot.  On success `*geom_out` (opaque to the caller) holds
// what the launch needs; called by the launch plans of gemm_f32.hip (launch_cfg<64, 64, A_M4, B_IM2COL*>) with that translation unit's GemmArgs.
int32_t rten_launch_gemm_f32_patch(rten_hip_ctx *ctx, const void *args, unsigned grid_x, unsigned grid_z, int mode) {
    GemmArgs a = *static_cast<const GemmArgs *>(args);
    if (!(a.KH == 3 && a.KW == 3 && a.sy == 1 && a.sx == 1 && a.dy == 1 && a.dx == 1 && a.pt == 1 && a.pl == 1 && a.OW == a.W && a.Pn == a.H * a.W && a.K % PK == 0 &&
          a.K >= PK && a.N > 0 && a.b_ns > 0))
        return RTEN_HIP_ERR_UNSUPPORTED;
    PatchGeom g;
    const bool x4 = (a.W & 3) == 0 && (a.b_ns & 3) == 0 && (((uintptr_t)a.B | (uintptr_t)(a.b_bs * 4) | (uintptr_t)(a.b_bsi * 4)) & 15) == 0;
    const int G = x4 ? 4 : 1;
    g.PW = x4 ? a.W + 4 : a.W + 1;
    g.HP = a.H + 2;
    g.chan_stride = a.H * a.W;
    g.n_images = (a.N + a.Pn - 1) / a.Pn;
    const long long vmax = (long long)g.n_images * g.HP * g.PW + 4 * g.PW + 512;
    if (vmax > 0x3fffffff) return RTEN_HIP_ERR_UNSUPPORTED;
    g.dPW = rten_make_div(vmax, g.PW);
    g.dHP = rten_make_div(vmax / g.PW + 2, g.HP);
    // the largest window any tile needs: from its first pixel's tap (0, 0), rounded down to a DMA lane, to its last pixel's tap (2, 2)
    auto pix_u = [&](long long n) {
        if (n > a.N - 1) n = a.N - 1;
        const long long nb = n / a.Pn, np = n - nb * a.Pn, oy = np / a.OW, ox = np - oy * a.OW;
        return (nb * g.HP + oy) * g.PW + ox - 1 + G;
    };
    long long ext = 0;
    for (long long n0 = 0; n0 < a.N; n0 += 64) {
        const long long w0 = pix_u(n0) / G * G, last = pix_u(n0 + 63) + 2 * g.PW + 2;
        if (last - w0 + 1 > ext) ext = last - w0 + 1;
    }
    const int np = (int)((ext + 64 * G - 1) / (64 * G));
    const dim3 grid(grid_x, grid_z);
    TRACE_ASSIGN(a, grid_x * grid_z);
    if (x4) {
        if (np <= 1) return launch_patch<4, 1>(ctx, a, g, grid, mode);
        if (np == 2) return launch_patch<4, 2>(ctx, a, g, grid, mode);
    } else {
        if (np <= 2) return launch_patch<1, 2>(ctx, a, g, grid, mode);
        if (np == 3) return launch_patch<1, 3>(ctx, a, g, grid, mode);
        if (np == 4) return launch_patch<1, 4>(ctx, a, g, grid, mode);
    }
    return RTEN_HIP_ERR_UNSUPPORTED;
}
