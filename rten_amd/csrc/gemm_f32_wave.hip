// One wave = one workgroup = one 64x64 output tile: the "wave tile" family of the f32 implicit-GEMM kernels (gfx950).
//
// Replaces the same reference code as gemm_f32.hip (rten-gemm/src/lib.rs:794-1093, kernels/simd_generic.rs:285-414, the virtual im2col of
// rten-gemm/src/im2col.rs:56-212 and conv_impl, src/ops/conv.rs:124-365) for the convolution operand layouts: A = prepacked k-major weights
// [K][M], B = dense two-level activations (1x1 / stride 1) or the im2col gather.
//
// Why a second family (round 4, tools/debug/f32_trace.py + tools/probes/wave_tile.hip, profiles/r07):
//   * In the 4-wave 64x64 kernels every wave owns ONE 32x32 accumulator block, so each v_mfma_f32_32x32x2_f32 depends on the wave's previous one,
//     and the workgroup synchronises (s_barrier) every 8 MFMAs.  The workgroup timeline of a whole ResNet-50 step shows what that costs: the
//     matrix pipe is 67-71 % busy INSIDE the k-loops (105-112 TF/s chip-equivalent while a k-loop runs), whereas the 128x128 variant -- four
//     independent blocks per wave -- runs at ~100 % in the same trace but quantises badly on ResNet's layer shapes.
//   * A one-wave workgroup keeps the 64x64 granularity AND four independent accumulator chains per wave (2 x 2 blocks: an MFMA never waits
//     for the previous one), needs no barrier at all (the LDS ring is private to the wave: a counted s_waitcnt vmcnt is the only
//     synchronisation) and halves the fragment traffic (2 + 2 ds_read_b32 feed 4 MFMAs).  Probe: 142 TF/s on a dense k-loop at two waves per
//     SIMD (0.93 of the pipe) against 130 for the 4-wave form.
//
// Numerics: exactly those of gemm_f32.hip (same fold_first / fold_next / store_out, same depth-block boundaries, same exact split-K with the
// last-arrival fold) -- bit-identical to the oracle; the variant sweeps in tests/ run this family next to the others.
#include "gemm_f32_common.h"

namespace {

// MODE 0: K <= 256 (one depth block); 1: several depth blocks folded in registers; 2: split-K producer (one K group of one split tile,
// raw accumulators parked in the slab, last arrival folds -- split_finish).
// BL: B_N4 (dense dwordx4), B_IM2COL_TAPS (<= 31 taps, per-lane padding mask) or B_IM2COL (general gather).
// BKW x NST: k-tile depth x LDS stages of the wave's private ring (16 x 2 = 16 KB, 8 x 4 = 16 KB, 16 x 3 = 24 KB per wave).
// TMW: the wave's tile is (32 TMW) x (32 TMW): 2 = 64x64 (four independent accumulator blocks), 1 = 32x32 -- ONE block per wave like the 4-wave
// kernels' waves, i.e. their granularity (a 64x64 tile = four free-running waves), but with a private operand ring per wave and therefore NO
// barrier: in the 4-wave kernels a wave that has finished its 8 MFMAs of a k-tile waits for its three siblings on the other SIMDs, and with 3-6
// workgroups interleaved per SIMD those waits line up into convoys (the in-loop matrix-pipe efficiency of every 64x64 pipeline is 67-71 %,
// whatever its MFMA shape or stage count; 128x128 tiles, four times fewer barriers per MFMA, run at ~100 %).  Price: each wave loads its own
// A and B tiles (twice the L2 -> LDS traffic and DMA instructions per FLOP).  Dense B only.
template <int BL, int MODE, int BKW, int NST, int TMW = 2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TMW == 2 ? 2 : 4, TMW == 2 ? 2 : 8))) void igemm_f32_wave_kernel(const GemmArgs p) {
    TR_DECL
    TR_STAMP(0)
    kernarg_prefetch<(int)sizeof(GemmArgs)>();
    constexpr bool MULTI_KC = MODE == 1, SPLIT = MODE == 2;
    static_assert(BL == B_N4 || BL == B_IM2COL || BL == B_IM2COL_TAPS, "wave tile kernel covers the conv operand layouts");
    constexpr int TM = TMW, TN = TMW, BMW = 32 * TMW, BNW = 32 * TMW;
    static_assert(TMW == 2 || (BL == B_N4 && MODE != 2), "32x32 wave tiles: dense B, no split-K form");
    constexpr int STAGE = BKW * (BMW + BNW); // floats per stage: A [BKW][64] then B [BKW][64]
    constexpr int NA = BKW * BMW / 256;      // dwordx4 DMA instructions per k-tile (A)
    constexpr int NBV = BKW * BNW / 256;     // dwordx4 (dense B)
    constexpr int NBG = BKW;                 // dword gathers (im2col B, 64-wide tiles): one per k row, 64 columns = 64 lanes
    constexpr int PER_TILE = NA + (BL == B_N4 ? NBV : NBG);
    constexpr int KCT = 256 / BKW;           // k-tiles per reference depth block (kc = 256)
    constexpr bool IM2COL = BL == B_IM2COL || BL == B_IM2COL_TAPS, TAPS = BL == B_IM2COL_TAPS;
    __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];

    const int lane = threadIdx.x;
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;

    int tile, grp = -1; // grp >= 0: this wave computes one K group of a split tile
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
    const int m0 = bm * BMW, n0 = bn * BNW;

    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    const float *Ab = p.A + (long long)zo * p.a_bs + (long long)zi * p.a_bsi;
    const float *Bb = p.B + (long long)zo * p.b_bs + (long long)zi * p.b_bsi;
    const long long c_zoff = (long long)zo * p.c_bs + (long long)zi * p.c_bsi;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, (int)p.b_bytes, 0x00020000);
    const int nk = (p.K + BKW - 1) / BKW;

    // ---- loop-invariant DMA source offsets: instruction j covers the flat tile range [j*256, j*256+256) floats (dwordx4: 4 k rows x 64)
    unsigned a_voff[NA];
#pragma unroll
    for (int j = 0; j < NA; j++) {
        const int f = j * 256 + lane * 4;
        const int k = f / BMW, m = m0 + f % BMW;
        // rows >= K lie past the end of the [K][M4] buffer (hardware range check); columns >= M4 must not wrap
        a_voff[j] = m < (int)p.a_cs ? (unsigned)(((long long)k * p.a_cs + m) * 4) : OOB;
    }
    const unsigned a_kstep = (unsigned)(BKW * p.a_cs * 4);

    [[maybe_unused]] unsigned b_voff[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] int b_krow[BL == B_N4 ? NBV : 1];
    [[maybe_unused]] unsigned b_kstep = 0;
    [[maybe_unused]] int im_iy0 = 0, im_ix0 = 0, im_pix = 0;
    [[maybe_unused]] unsigned im_inv = 0; // TAPS: bit t set = tap t of this lane's pixel is padding; bit 31 always set (k-tail rows)
    if constexpr (BL == B_N4) {
#pragma unroll
        for (int j = 0; j < NBV; j++) {
            const int f = j * 256 + lane * 4;
            const int k = f / BNW, n = n0 + f % BNW;
            const int nn = n < p.N ? n : 0;
            const int nb = nn / p.Pn, np = nn - nb * p.Pn;
            b_krow[j] = k;
            b_voff[j] = n < p.N ? (unsigned)(((long long)k * p.b_rs + (long long)nb * p.b_ns + np) * 4) : OOB;
        }
        b_kstep = (unsigned)(BKW * p.b_rs * 4);
    } else {
        const int n = n0 + lane; // a gather instruction covers one k row x 64 columns: a lane sees ONE column
        const bool ok = n < p.N;
        const int nn = ok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const int oy = np / p.OW, ox = np - oy * p.OW;
        im_iy0 = ok ? oy * p.sy - p.pt : -0x40000000;
        im_ix0 = ox * p.sx - p.pl;
        im_pix = (int)((long long)nb * p.b_ns) + (oy * p.sy - p.pt) * p.W + im_ix0;
        if constexpr (TAPS) {
            unsigned colbad = 0; // bit kx set: column tap kx falls outside the image
            for (int kx = 0; kx < p.KW; kx++) colbad |= ((unsigned)(im_ix0 + kx * p.dx) >= (unsigned)p.W ? 1u : 0u) << kx;
            const unsigned allbad = (1u << p.KW) - 1u;
            unsigned inv = 0x80000000u;
            for (int ky = 0; ky < p.KH; ky++) inv |= ((unsigned)(im_iy0 + ky * p.dy) >= (unsigned)p.H ? allbad : colbad) << (ky * p.KW);
            im_inv = inv;
        }
    }

    // im2col LUT entries (scalar loads) for the tile whose DMA is issued NEXT: all BKW rows belong to this wave
    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    [[maybe_unused]] i32x2 lutE[IM2COL ? BKW : 1];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (IM2COL) {
            const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)p.lut;
#pragma unroll
            for (int j = 0; j < BKW; j++) lutE[j] = lc[kt * BKW + j];
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        float *Bs = As + BKW * BMW;
        const int kts = kt < nk ? kt : (nk > 0 ? nk - 1 : 0); // keep the scalar offset inside the buffer
        const bool past = kt >= nk;
        const unsigned a_soff = (unsigned)kts * a_kstep;
#pragma unroll
        for (int j = 0; j < NA; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + j * 256), 16, (int)(past ? OOB : a_voff[j]), (int)a_soff, 0, 0);
        if constexpr (BL == B_N4) {
            const int kleft = p.K - kt * BKW;
            const unsigned b_soff = (unsigned)kts * b_kstep;
#pragma unroll
            for (int j = 0; j < NBV; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + j * 256), 16, (int)(b_krow[j] < kleft ? b_voff[j] : OOB), (int)b_soff, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NBG; r++) {
                const i32x2 e = lutE[r];
                unsigned voff;
                if constexpr (TAPS) {
                    // e[1] = 31 - tap: the tap's padding bit moves to bit 31 and pushes the offset out of range
                    voff = ((im_inv << e[1]) & 0x80000000u) | ((unsigned)(im_pix + e[0]) << 2);
                } else {
                    const int iy = im_iy0 + (e[1] & 0xffff);
                    const int ix = im_ix0 + (e[1] >> 16);
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    voff = ok ? (unsigned)(im_pix + e[0]) << 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + r * BNW), 4, (int)voff, 0, 0, 0);
            }
        }
    };

    // ---- accumulators: 2 x 2 independent 32x32 blocks; `tot` holds the folded depth blocks (MULTI_KC)
    f32x16 acc[TM][TN];
    [[maybe_unused]] f32x16 tot[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };
    zero_acc();
    [[maybe_unused]] auto flush = [&](bool first) {
        int mb = m0 + 4 * half, nb0 = n0 + l31;
        asm volatile("" : "+v"(mb), "+v"(nb0));
        if (first) fold_first<TM, TN>(p, z, acc, tot, mb, nb0, c_zoff);
        else fold_next<TM, TN>(p, acc, tot);
        zero_acc();
    };

    auto compute_tile = [&](int stage) {
        // A fragment of k-pair kk, block i: As[2*kk + half][i*32 + l31]; B fragment, block j: Bs[2*kk + half][j*32 + l31]
        const float *As = smem + stage * STAGE + l31 + half * BMW;
        const float *Bs = smem + stage * STAGE + BKW * BMW + l31 + half * BNW;
        float af[2][TM], bf[2][TN]; // double buffered across k-pairs: the next pair's ds_reads sit behind the current MFMA group
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = As[i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = Bs[j * 32];
#pragma unroll
        for (int kk = 0; kk < BKW / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKW / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[nxt][i] = As[2 * (kk + 1) * BMW + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[2 * (kk + 1) * BNW + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
    };

    // raw accumulator image of this tile in the split-K slab: [i][j][quad][lane] float4 (= the 4-wave kernels' [wave][quad][lane] image)
    [[maybe_unused]] auto store_raw = [&](f32x16 (&v)[TM][TN], int slot) {
        int loff = lane * 4;
        asm volatile("" : "+v"(loff)); // keep the address math at the use (not hoisted across the K loop)
        float *base = p.slab + (((long long)z * p.split_ntail + (tile - p.split_t1)) * p.split_slots + slot) * (long long)(BMW * BNW) + loff;
        if (p.split_counters) { // folded in this launch, possibly on another XCD: write through (see coherent_store4)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base - loff), 0, BMW * BNW * 4, 0x00020000);
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

    // ---- software pipeline: NST - 1 k-tiles in flight behind the one being multiplied.  No barrier anywhere: the ring is this wave's own,
    // a stage is refilled only after the MFMAs that read it have been issued (in order) and their fragments consumed.
    const int nblk = (MULTI_KC || SPLIT) ? (nk + KCT - 1) / KCT : 1;
    int blk0 = 0, blk1 = nblk;
    if constexpr (SPLIT) {
        blk0 = grp * p.split_g;
        blk1 = blk0 + p.split_g < nblk ? blk0 + p.split_g : nblk;
    }
    const int kt0 = blk0 * KCT;
    fetch_lut(kt0);
#pragma unroll
    for (int i = 0; i < NST - 1; i++) {
        issue_tile(kt0 + i, i);
        fetch_lut(kt0 + i + 1);
    }
    int stage = 0;
    TR_STAMP(1)
    for (int blk = blk0; blk < blk1; blk++) {
        const int kt_end = (MULTI_KC || SPLIT) ? ((blk + 1) * KCT < nk ? (blk + 1) * KCT : nk) : nk;
        for (int kt = blk * KCT; kt < kt_end; kt++) {
            wait_vmcnt<PER_TILE *(NST - 2)>(); // this wave's DMA for tile kt has landed; NST-2 later tiles stay in flight
#ifdef RTEN_TRACE
            if (tr_trips == 0) TR_STAMP(2)
            tr_trips++;
#endif
            const int stp = stage == 0 ? NST - 1 : stage - 1; // the stage tile kt-1 used
            issue_tile(kt + NST - 1, stp);
            fetch_lut(kt + NST);
            compute_tile(stage);
            stage = stage == NST - 1 ? 0 : stage + 1;
        }
        if constexpr (SPLIT) {
            store_raw(acc, blk);
            zero_acc();
        } else if constexpr (MULTI_KC) {
            if (blk + 1 < nblk) flush(blk == 0);
        }
    }
    wait_vmcnt<0>(); // drain the look-ahead tiles before the LDS goes away
    TR_STAMP(3)
    [[maybe_unused]] constexpr unsigned TR_KID = (unsigned)BMW | ((unsigned)BNW << 8) | (MODE << 16) | (BL << 20) | (3u << 24) | ((unsigned)NST << 28);

    if constexpr (SPLIT) {
        if (p.split_counters) split_finish<BMW, BNW, TM, TN, 1, 1>(p, z, tile, 0, lane, m0, n0, c_zoff, reinterpret_cast<int *>(smem));
        TR_STAMP(4)
        TR_WRITE(TR_KID, tile, grp)
        return;
    } else {
        const int mb = m0 + 4 * half, nb0 = n0 + l31;
        if constexpr (MULTI_KC) { // launched only for K > 256: at least two depth blocks
            fold_next<TM, TN>(p, acc, tot);
            store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
        } else {
            fold_first<TM, TN>(p, z, acc, acc, mb, nb0, c_zoff);
            store_out<TM, TN>(p, acc, mb, nb0, c_zoff);
        }
        TR_STAMP(4)
        TR_WRITE(TR_KID, tile, grp)
    }
}

template <int BL, int MODE>
int32_t launch_flavour(rten_hip_ctx *ctx, const GemmArgs &a, dim3 grid, int flavour) {
    if constexpr (BL == B_N4 && MODE != 2) {
        if (flavour >= 4) { // 32x32 wave tiles: the caller's grid counts 64x64 tiles -- recount
            GemmArgs b = a;
            b.tiles_m = (a.M + 31) / 32;
            b.tiles_n = (a.N + 31) / 32;
            const dim3 g((unsigned)(b.tiles_m * b.tiles_n), grid.y);
            TRACE_ASSIGN(b, g.x * g.y);
            if (flavour == 5) hipLaunchKernelGGL((igemm_f32_wave_kernel<BL, MODE, 16, 3, 1>), g, dim3(64), 0, ctx->stream, b);
            else hipLaunchKernelGGL((igemm_f32_wave_kernel<BL, MODE, 16, 2, 1>), g, dim3(64), 0, ctx->stream, b);
            RTEN_LAUNCH_CHECK(ctx, "igemm_f32_wave_kernel (32x32) launch");
            return RTEN_HIP_OK;
        }
    }
    switch (flavour) {
    case 1: hipLaunchKernelGGL((igemm_f32_wave_kernel<BL, MODE, 8, 4>), grid, dim3(64), 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((igemm_f32_wave_kernel<BL, MODE, 16, 3>), grid, dim3(64), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((igemm_f32_wave_kernel<BL, MODE, 16, 2>), grid, dim3(64), 0, ctx->stream, a); break;
    }
    RTEN_LAUNCH_CHECK(ctx, "igemm_f32_wave_kernel launch");
    return RTEN_HIP_OK;
}

template <int BL>
int32_t launch_mode(rten_hip_ctx *ctx, const GemmArgs &a, dim3 grid, int mode, int flavour) {
    switch (mode) {
    case 2: return launch_flavour<BL, 2>(ctx, a, grid, flavour);
    case 1: return launch_flavour<BL, 1>(ctx, a, grid, flavour);
    default: return launch_flavour<BL, 0>(ctx, a, grid, flavour);
    }
}

} // namespace

// Called by the launch plans of gemm_f32.hip (launch_cfg<64, 64, A_M4, BL>): `args` is that translation unit's GemmArgs (same header, same layout).
int32_t rten_launch_gemm_f32_wave(rten_hip_ctx *ctx, const void *args, unsigned grid_x, unsigned grid_z, int bl, int mode, int flavour) {
    GemmArgs a = *static_cast<const GemmArgs *>(args);
    const dim3 grid(grid_x, grid_z);
    TRACE_ASSIGN(a, grid_x * grid_z);
    switch (bl) {
    case B_N4: return launch_mode<B_N4>(ctx, a, grid, mode, flavour);
    case B_IM2COL_TAPS: return launch_mode<B_IM2COL_TAPS>(ctx, a, grid, mode, flavour);
    case B_IM2COL: return launch_mode<B_IM2COL>(ctx, a, grid, mode, flavour);
    default: return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "internal: wave tile kernel called with an operand layout it does not cover");
    }
}
