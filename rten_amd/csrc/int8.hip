// MFMA int8 GEMM / implicit-GEMM convolution for gfx950: v_mfma_i32_32x32x32_i8.
//
// Replaces: the u8 x i8 -> i32 micro-kernel + zero-point epilogue (rten-gemm/src/kernels/generic.rs:274-366,
// simd_generic.rs:576-780), int8 packing incl. row/column sums (packing/int8.rs), the int8 im2col
// packing (im2col.rs:264-389) and the operator front-ends matmul_integer / MatMulIntegerToFloat
// (src/ops/matmul.rs:582-647,789-794), conv_integer / ConvIntegerToFloat (src/ops/conv.rs:421-476,
// 571-578), cast_scale (matmul.rs:734-773) and ShiftCast (src/shift_cast.rs:39-50).
//
//     C[m,n] = sum_k (A[m,k] - a_zp[m]) * (B[k,n] - b_zp[n])                      (wrapping i32)
//
// MI355X mapping.  MFMA needs both operands signed, so u8 operands are moved to the signed domain on
// the way into LDS (x ^ 0x80 == x - 128) together with their zero points; (x - zp) is invariant under
// that shift, so the result is the reference's, bit for bit, and the reference's two full-tensor
// shift-cast copies (conv.rs:458-461) disappear.  The zero-point algebra is the reference's
// (simd_generic.rs:676-746):  C = dot - b_zp*rowsum(A) - a_zp*colsum(B) + K*a_zp*b_zp, with the row and
// column sums accumulated by the loader threads (v_dot4 against 0x01010101) while they stage the tiles.
// LDS tiles are k-contiguous ([64 rows][64 B + 16 B pad]) so an MFMA operand is one conflict-free
// ds_read_b128 per lane.  Epilogue fuses cast_scale, the bias Add, residual Add and Relu.
//
// This first version gathers NCHW bytes (4 byte loads per LDS dword); the NHWC-i8 activation layout that
// turns the gather into 16 B loads is the next step ("int8 roadmap").
#include "internal.h"
#include "vecmath.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// int8_fast.hip: k-contiguous staging + LDS-DMA MFMA kernel; RTEN_HIP_ERR_UNSUPPORTED = not covered, use the generic kernel
int32_t rten_i8_fast_gemm(rten_hip_ctx *ctx, const rten_hip_gemm_int8_desc *d, const void *a, const void *b, const void *a_zp,
                          const void *b_zp, const float *scale, void *c);
int32_t rten_i8_fast_conv(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp,
                          const void *w_zp, const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats);

namespace {

constexpr int IBM = 64, IBN = 64, IBK = 64;
constexpr int LDS_ROW = IBK + 16; // bytes
constexpr int NT = 256;

struct I8Args {
    const uint8_t *A;
    const uint8_t *B;
    void *C;
    const uint8_t *a_zp;
    const uint8_t *b_zp;
    const float *scale;
    const float *bias;
    const float *res;
    int M, N, K;
    long long a_rs, a_cs, a_bs;
    long long b_rs, b_cs, b_ns, b_bs;
    long long c_rs, c_ns, c_bs;
    int Pn;
    int a_signed, b_signed;
    int a_zp_len, b_zp_len, a_zp_bs; // a_zp index = z*a_zp_bs + (len==1 ? 0 : m)
    int scale_len;
    int scale_per_row; // conv: per-output-channel scale
    int bias_bs;
    int relu;
    int tiles_m, tiles_n;
    // im2col
    int im2col;
    int H, W, HW, KHW, KW, OW, sy, sx, dy, dx, pt, pl;
    unsigned magic_khw, magic_kw;
    int pad_mode;
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

__device__ __forceinline__ int zp_signed(const uint8_t *zp, int idx, int is_signed) {
    // value of the zero point in the signed (MFMA) domain
    if (!zp) return is_signed ? 0 : -128;
    return is_signed ? (int)(int8_t)zp[idx] : (int)zp[idx] - 128;
}

template <bool IM2COL>
__global__ __launch_bounds__(NT, 2) void igemm_i8_kernel(const I8Args p) {
    __shared__ __attribute__((aligned(16))) uint8_t As[IBM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) uint8_t Bs[IBN * LDS_ROW];
    __shared__ int rsum[IBM], csum[IBN];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;
    const int bm = blockIdx.x % p.tiles_m, bn = blockIdx.x / p.tiles_m;
    const int m0 = bm * IBM, n0 = bn * IBN;
    const uint8_t *__restrict__ Ab = p.A + (long long)z * p.a_bs;
    const uint8_t *__restrict__ Bb = p.B + (long long)z * p.b_bs;
    const unsigned a_flip = p.a_signed ? 0u : 0x80808080u;
    const unsigned b_flip = p.b_signed ? 0u : 0x80808080u;

    // loader mapping: dword index idx = t + j*256 -> (row = idx / 16, kq = idx % 16); rows fixed per thread
    const int kq = t & 15;
    long long a_off[4], b_off[4];
    bool a_ok[4], b_ok[4];
    int im_iy0[4], im_ix0[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = (t >> 4) + 16 * j;
        const int m = m0 + row;
        a_ok[j] = m < p.M;
        a_off[j] = (long long)(a_ok[j] ? m : 0) * p.a_rs;
        const int n = n0 + row;
        b_ok[j] = n < p.N;
        const int nn = b_ok[j] ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        if constexpr (IM2COL) {
            const int oy = np / p.OW, ox = np - oy * p.OW;
            im_iy0[j] = oy * p.sy - p.pt;
            im_ix0[j] = ox * p.sx - p.pl;
            b_off[j] = (long long)nb * p.b_ns;
        } else {
            b_off[j] = (long long)nb * p.b_ns + (long long)np * p.b_cs;
        }
    }
    // padded taps in the signed domain
    int pad_s = 0;
    if constexpr (IM2COL) {
        if (p.pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = zp_signed(p.b_zp, 0, p.b_signed);
        else if (p.pad_mode == RTEN_HIP_PAD_RAW0_I8) pad_s = 0;
        else pad_s = -128;
    }

    unsigned ra[4], rb[4];
    int rs_part[4] = {0, 0, 0, 0}, cs_part[4] = {0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        const int k0 = kt * IBK + kq * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int k = k0 + b;
                const bool ok = a_ok[j] && k < p.K;
                const unsigned v = Ab[ok ? a_off[j] + (long long)k * p.a_cs : 0ll];
                w |= (ok ? ((v ^ (a_flip & 0xffu)) & 0xffu) : 0u) << (8 * b);
            }
            ra[j] = w;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int k = k0 + b;
                unsigned byte;
                if constexpr (IM2COL) {
                    const unsigned c = fastdiv((unsigned)k, p.magic_khw);
                    const unsigned rem = (unsigned)k - c * (unsigned)p.KHW;
                    const unsigned ky = fastdiv(rem, p.magic_kw);
                    const unsigned kx = rem - ky * (unsigned)p.KW;
                    const int iy = im_iy0[j] + (int)ky * p.dy, ix = im_ix0[j] + (int)kx * p.dx;
                    const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const bool kok = b_ok[j] && k < p.K;
                    const unsigned v = Bb[(kok && inb) ? b_off[j] + (long long)c * p.HW + (long long)iy * p.W + ix : 0ll];
                    byte = !kok ? 0u : (inb ? ((v ^ (b_flip & 0xffu)) & 0xffu) : ((unsigned)pad_s & 0xffu));
                } else {
                    const bool ok = b_ok[j] && k < p.K;
                    const unsigned v = Bb[ok ? b_off[j] + (long long)k * p.b_rs : 0ll];
                    byte = ok ? ((v ^ (b_flip & 0xffu)) & 0xffu) : 0u;
                }
                w |= byte << (8 * b);
            }
            rb[j] = w;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = (t >> 4) + 16 * j;
            *reinterpret_cast<unsigned *>(As + row * LDS_ROW + kq * 4) = ra[j];
            *reinterpret_cast<unsigned *>(Bs + row * LDS_ROW + kq * 4) = rb[j];
            rs_part[j] = __builtin_amdgcn_sdot4((int)ra[j], 0x01010101, rs_part[j], false);
            cs_part[j] = __builtin_amdgcn_sdot4((int)rb[j], 0x01010101, cs_part[j], false);
        }
    };

    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    i32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0;

    const int nk = (p.K + IBK - 1) / IBK;
    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; kt++) {
        __syncthreads(); // previous tile's MFMA reads are done
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1); // global loads in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < IBK / 32; kk++) {
            const i32x4 af = *reinterpret_cast<const i32x4 *>(As + (wm0 + l31) * LDS_ROW + kk * 32 + half * 16);
            const i32x4 bf = *reinterpret_cast<const i32x4 *>(Bs + (wn0 + l31) * LDS_ROW + kk * 32 + half * 16);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc, 0, 0, 0);
        }
    }

    // row / column sums: reduce over the 16 loader threads that share a row
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int r = rs_part[j], c = cs_part[j];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            r += __shfl_xor(r, o, 64);
            c += __shfl_xor(c, o, 64);
        }
        if (kq == 0) {
            rsum[(t >> 4) + 16 * j] = r;
            csum[(t >> 4) + 16 * j] = c;
        }
    }
    __syncthreads();

    // epilogue
    const int n = n0 + wn0 + l31;
    if (n >= p.N) return;
    const int nb = n / p.Pn, np = n - nb * p.Pn;
    const long long ccol = (long long)z * p.c_bs + (long long)nb * p.c_ns + np;
    const unsigned bz = (unsigned)zp_signed(p.b_zp, p.b_zp_len == 1 ? 0 : n, p.b_signed);
    const unsigned cs = (unsigned)csum[wn0 + l31];
    const float sc = (p.scale && !p.scale_per_row) ? p.scale[p.scale_len == 1 ? 0 : n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ml = wm0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + ml;
        if (m >= p.M) continue;
        const unsigned az = (unsigned)zp_signed(p.a_zp, z * p.a_zp_bs + (p.a_zp_len > 1 ? m % p.a_zp_len : 0), p.a_signed); // period < M: cycled zero points (matmul.rs:266-280)
        const unsigned v = (unsigned)acc[r] - bz * (unsigned)rsum[ml] - az * cs + (unsigned)p.K * az * bz;
        const long long off = ccol + (long long)m * p.c_rs;
        if (p.scale) {
            float f = (float)(int)v * (p.scale_per_row ? p.scale[z * p.bias_bs + m] : sc); // cast_scale (matmul.rs:751,761)
            if (p.bias) f = f + p.bias[z * p.bias_bs + m];   // following Add(bias [1,O,1,1])
            if (p.res) f = f + p.res[off];                   // following residual Add
            if (p.relu) f = vm::relu(f);
            reinterpret_cast<float *>(p.C)[off] = f;
        } else {
            reinterpret_cast<int *>(p.C)[off] = (int)v;
        }
    }
}

unsigned magic_for(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

int32_t launch_i8(rten_hip_ctx *ctx, I8Args &a, int Z) {
    a.tiles_m = (a.M + IBM - 1) / IBM;
    a.tiles_n = (a.N + IBN - 1) / IBN;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)Z);
    const double ops = 2.0 * a.M * (double)a.N * a.K * Z;
    const double bytes = (double)Z * ((double)a.M * a.K + (double)a.K * a.N + 4.0 * a.M * a.N);
    if (a.im2col) {
        ProfScope ps(ctx, "igemm_i8_kernel<1>", ops, bytes);
        hipLaunchKernelGGL((igemm_i8_kernel<true>), grid, dim3(NT), 0, ctx->stream, a);
    } else {
        ProfScope ps(ctx, "igemm_i8_kernel<0>", ops, bytes);
        hipLaunchKernelGGL((igemm_i8_kernel<false>), grid, dim3(NT), 0, ctx->stream, a);
    }
    RTEN_LAUNCH_CHECK(ctx, "igemm_i8_kernel launch");
    return RTEN_HIP_OK;
}

} // namespace

namespace {
// One product of the (possibly batched) call.
int32_t gemm_int8_one(rten_hip_ctx *ctx, const rten_hip_gemm_int8_desc *d, const void *a, const void *b, const void *a_zp, const void *b_zp,
                      const float *scale, void *c) {
    if ((ctx->int8_path == 0 || d->b_prepacked) && d->k > 0) {
        const int32_t rc = rten_i8_fast_gemm(ctx, d, a, b, a_zp, b_zp, scale, c);
        if (rc != RTEN_HIP_ERR_UNSUPPORTED) return rc;
        if (d->b_prepacked) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: prepacked RHS for a shape the staged kernel does not cover (packed_bytes == 0)");
    }
    I8Args g = {};
    g.A = (const uint8_t *)a; g.B = (const uint8_t *)b; g.C = c;
    g.a_zp = d->a_zp_len ? (const uint8_t *)a_zp : nullptr;
    g.b_zp = d->b_zp_len ? (const uint8_t *)b_zp : nullptr;
    g.scale = d->scale_len ? scale : nullptr;
    g.M = d->m; g.N = d->n; g.K = d->k;
    g.a_rs = d->a_rs; g.a_cs = d->a_cs; g.b_rs = d->b_rs; g.b_cs = d->b_cs;
    g.c_rs = d->ldc; g.Pn = d->n;
    g.a_signed = d->a_signed; g.b_signed = d->b_signed;
    g.a_zp_len = d->a_zp_len; g.b_zp_len = d->b_zp_len; g.scale_len = d->scale_len;
    return launch_i8(ctx, g, 1);
}
} // namespace

RTEN_EXPORT int32_t rten_hip_gemm_int8(rten_hip_ctx *ctx, const rten_hip_gemm_int8_desc *d, const void *a,
                                       const void *b, const void *a_zp, const void *b_zp, const float *scale,
                                       void *c) {
    RTEN_CHECK_CTX(ctx);
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: negative dimension");
    const int batch = d->batch <= 1 ? 1 : d->batch;
    if (d->m == 0 || d->n == 0) return RTEN_HIP_OK;
    if (!c || (d->k > 0 && (!a || !b))) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: NULL operand");
    // a_zp: scalar, per row, or a period dividing m (cycled zero points of a collapsed batched LHS, matmul.rs:266-280)
    if ((d->a_zp_len < 0 || (d->a_zp_len > 1 && d->m % d->a_zp_len != 0)) || (d->b_zp_len != 0 && d->b_zp_len != 1 && d->b_zp_len != d->n))
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Zero point has incorrect size"); // matmul.rs:523
    if ((d->a_zp_len && !a_zp) || (d->b_zp_len && !b_zp)) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: zero point length set but pointer is NULL");
    if (d->scale_len != 0 && d->scale_len != 1 && d->scale_len != d->n)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "Scale length does not match tensor columns");
    if (d->scale_len && !scale) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: scale_len set but scale is NULL");
    if (d->b_prepacked && batch > 1 && d->b_bs != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "gemm_int8: a prepacked RHS is a single matrix (b_bs must be 0)");
    // batched_gemm_uninit (matmul.rs:302-372): independent products over the broadcast prefix, same quantization
    // parameters for each; the staging buffers are reused in stream order.
    const size_t csz = d->scale_len ? sizeof(float) : sizeof(int32_t);
    for (int z = 0; z < batch; z++) {
        const int32_t rc = gemm_int8_one(ctx, d, (const uint8_t *)a + (long long)z * d->a_bs, d->b_prepacked ? b : (const void *)((const uint8_t *)b + (long long)z * d->b_bs), a_zp, b_zp,
                                         scale, (uint8_t *)c + (size_t)z * (size_t)d->c_bs * csz);
        if (rc) return rc;
    }
    return RTEN_HIP_OK;
}

namespace {
int32_t conv2d_int8_impl(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp, const void *w_zp,
                         const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats);
}

RTEN_EXPORT int32_t rten_hip_conv2d_int8(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x,
                                         const void *w, const void *x_zp, const void *w_zp, const float *scale,
                                         const float *bias, const float *residual, uint32_t flags, void *y) {
    return conv2d_int8_impl(ctx, di, x, w, x_zp, w_zp, scale, bias, residual, flags, y, nullptr);
}

// Same, and the float outputs' min/max are accumulated into `stats` (rten_hip_minmax_stats_reset first) for the
// DynamicQuantizeLinear that quantizes them next: that operator's first sweep disappears.  Needs the staged kernel
// (RTEN_HIP_ERR_UNSUPPORTED otherwise: call rten_hip_conv2d_int8 and quantize with the two-sweep form).
RTEN_EXPORT int32_t rten_hip_conv2d_int8_stats(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x,
                                               const void *w, const void *x_zp, const void *w_zp, const float *scale,
                                               const float *bias, const float *residual, uint32_t flags, void *y, void *stats) {
    if (!stats || !scale) return RTEN_HIP_ERR_INVALID_VALUE;
    return conv2d_int8_impl(ctx, di, x, w, x_zp, w_zp, scale, bias, residual, flags, y, stats);
}

namespace {
int32_t conv2d_int8_impl(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp, const void *w_zp,
                         const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats) {
    RTEN_CHECK_CTX(ctx);
    if (!di) return RTEN_HIP_ERR_INVALID_VALUE;
    const rten_hip_conv2d_desc *d = &di->conv;
    if (d->groups <= 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Group count must be > 0");
    if (d->c % d->groups != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Input channel count not divisible by groups");
    if (d->o % d->groups != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Output channel count not divisible by groups");
    if (di->w_zp_len != 0 && di->w_zp_len != 1 && di->w_zp_len != d->o)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Zero point has incorrect size");
    if (!x || !w || !y) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv_int8: NULL operand");
    if (!scale && (bias || residual || (flags & RTEN_HIP_CONV_RELU)))
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv_int8: float epilogue needs a scale");
    if (di->scale_len != 0 && di->scale_len != 1 && di->scale_len != d->o)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "conv_int8: scale must be a scalar or have one value per output channel");
    if (d->n == 0 || d->out_h == 0 || d->out_w == 0) return RTEN_HIP_OK;
    const int Cg = d->c / d->groups, Og = d->o / d->groups;
    const int K = Cg * d->kh * d->kw, P = d->out_h * d->out_w;
    const long long HW = (long long)d->h * d->w;
    if ((long long)d->n * d->c * HW >= (1ll << 31) || (long long)d->n * d->o * P >= (1ll << 31))
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv: tensors above 2^31 elements are not supported");
    if (ctx->int8_path == 0 || di->weights_packed || di->x_staged || stats) {
        const int32_t rc = rten_i8_fast_conv(ctx, di, x, w, x_zp, w_zp, scale, bias, residual, flags, y, stats);
        if (rc != RTEN_HIP_ERR_UNSUPPORTED) return rc;
    }
    if (stats) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv_int8_stats: geometry not covered by the staged kernel");
    I8Args g = {};
    // kernel is the LHS (conv.rs:461-474): A = W[o][k], B = im2col(x)
    g.A = (const uint8_t *)w; g.B = (const uint8_t *)x; g.C = y;
    g.a_zp = di->w_zp_len ? (const uint8_t *)w_zp : nullptr;
    g.b_zp = (const uint8_t *)x_zp; // scalar or NULL
    g.scale = scale; g.bias = bias;
    g.res = (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr;
    g.relu = (flags & RTEN_HIP_CONV_RELU) ? 1 : 0;
    g.M = Og; g.N = d->n * P; g.K = K;
    g.a_rs = K; g.a_cs = 1; g.a_bs = (long long)Og * K;
    g.b_bs = (long long)Cg * HW; g.b_ns = (long long)d->c * HW;
    g.c_rs = P; g.c_ns = (long long)d->o * P; g.c_bs = (long long)Og * P;
    g.Pn = P;
    g.a_signed = di->w_signed; g.b_signed = di->x_signed;
    g.a_zp_len = di->w_zp_len; g.a_zp_bs = di->w_zp_len > 1 ? Og : 0;
    g.b_zp_len = x_zp ? 1 : 0;
    g.scale_len = scale ? 1 : 0;
    g.scale_per_row = (scale && di->scale_len > 1) ? 1 : 0;
    g.bias_bs = Og;
    g.im2col = 1;
    g.H = d->h; g.W = d->w; g.HW = (int)HW; g.KHW = d->kh * d->kw; g.KW = d->kw; g.OW = d->out_w;
    g.sy = d->stride_h; g.sx = d->stride_w; g.dy = d->dil_h; g.dx = d->dil_w;
    g.pt = d->pads[0]; g.pl = d->pads[1];
    g.magic_khw = magic_for((unsigned)g.KHW); g.magic_kw = magic_for((unsigned)g.KW);
    g.pad_mode = rten_effective_pad_mode(di);
    return launch_i8(ctx, g, d->groups);
}
} // namespace

// 0 = automatic (k-contiguous staging + LDS-DMA kernel whenever it covers the call), 1 = generic kernel only.
RTEN_EXPORT int32_t rten_hip_set_int8_path(rten_hip_ctx *ctx, int32_t mode) {
    RTEN_CHECK_CTX(ctx);
    if (mode < 0 || mode > 1) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "set_int8_path: mode must be 0 or 1");
    ctx->int8_path = mode;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_set_int8_tile(rten_hip_ctx *ctx, int32_t tile, int32_t *previous) {
    RTEN_CHECK_CTX(ctx);
    if (tile < -1 || tile > 3) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "set_int8_tile: tile must be -1 (per-shape rule), 0 (128x128), 1 (128x64), 2 (64x128) or 3 (64x64)");
    if (previous) *previous = ctx->int8_tile;
    ctx->int8_tile = tile;
    return RTEN_HIP_OK;
}
