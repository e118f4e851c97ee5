"""The addressing scheme of the image-patch convolution kernels (rten_amd/csrc/gemm_f32_patch.hip), restated in numpy and checked against a plain padded
im2col -- no GPU.  The kernel stages, per channel, a window of a VIRTUAL zero-padded image stack ([image][H + 2 rows][PW dwords]) under a tile's 64 output
pixels and lets lane (pixel, tap) read window[base(pixel) + ky * PW + kx].  What is pinned here: (1) every (pixel, tap) read lands on the input element the
reference's im2col names (rten-gemm/src/im2col.rs:56-212) or on a zero, for ragged sizes, several images and tiles that straddle rows and images; (2) the
window the host sizes (`ext` in rten_launch_gemm_f32_patch) covers every read; (3) a DMA lane on the dwordx4 path never mixes data and padding."""
import numpy as np
import pytest


def window_model(x, n0, G):
    """x: [images, H, W] (one channel).  Returns (window as the DMA would fill it, per-pixel base offsets, PW) for the tile starting at pixel n0."""
    I, H, W = x.shape
    PW = W + 4 if G == 4 else W + 1
    HP = H + 2
    N, Pn = I * H * W, H * W

    def pix_u(n):
        n = min(n, N - 1)
        nb, np_ = divmod(n, Pn)
        oy, ox = divmod(np_, W)
        return (nb * HP + oy) * PW + ox - 1 + G

    w0 = pix_u(n0) // G * G
    last = pix_u(n0 + 63) + 2 * PW + 2
    ext = last - w0 + 1
    lanes = -(-ext // G)
    win = np.zeros(lanes * G, np.float32)
    for lane in range(lanes):
        u = w0 + lane * G - G  # unshifted virtual dword of the lane's first element
        if u < 0:
            continue
        v, col = divmod(u, PW)
        img, r = divmod(v, HP)
        ok = col < W and 1 <= r <= H and img < I
        if G == 4:
            assert col % 4 == 0 and (not ok or col + 3 < W)  # (3): a lane is all data or all padding
        if ok:
            win[lane * G:(lane + 1) * G] = x[img, r - 1, col:col + G]
    bases = [pix_u(n0 + j) - w0 for j in range(64)]
    return win, bases, PW, ext


@pytest.mark.parametrize("images,H,W", [(2, 56, 56), (3, 28, 28), (5, 14, 14), (9, 7, 7), (2, 9, 11), (1, 5, 3), (4, 6, 8), (1, 1, 1), (3, 2, 70)])
def test_every_tap_of_every_pixel_reads_its_im2col_element(images, H, W):
    rng = np.random.default_rng(images * 1000 + H * 10 + W)
    x = rng.standard_normal((images, H, W)).astype(np.float32)
    xp = np.zeros((images, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    N = images * H * W
    for G in ((4, 1) if W % 4 == 0 else (1,)):
        worst = 0
        for n0 in range(0, N, 64):
            win, bases, PW, ext = window_model(x, n0, G)
            worst = max(worst, ext)
            for j in range(64):
                n = min(n0 + j, N - 1)  # columns past N are clamped (their results are dropped by the store)
                img, rem = divmod(n, H * W)
                oy, ox = divmod(rem, W)
                for ky in range(3):
                    for kx in range(3):
                        idx = bases[j] + ky * PW + kx
                        assert 0 <= idx < ext <= len(win) + G, (n0, j, ky, kx)
                        assert win[idx] == xp[img, oy + ky, ox + kx], (G, n0, j, ky, kx)
        # the slot sizes the kernels are instantiated with: NP * 64 lanes of G dwords (dwordx4: NP <= 2, dword: NP <= 4) -- larger windows are refused
        np_needed = -(-worst // (64 * G))
        if (images, H, W) in ((2, 56, 56), (3, 28, 28), (5, 14, 14), (9, 7, 7)):
            assert np_needed <= (2 if G == 4 else 4), (G, worst)  # every ResNet-50 3x3 geometry is covered


def test_k_tiles_of_two_channels_and_where_the_depth_blocks_end():
    """k = c * 9 + ky * 3 + kx; a k-tile is 18 consecutive k (two channels); the reference's depth blocks end at multiples of 256 (rten-gemm/src/lib.rs:630-633),
    always between two k-pairs, inside a tile: the kernel's `interior` test must call exactly the other tiles interior."""
    for K in (18, 144, 576, 1152, 2304, 4608):
        for k_begin, k_end in ((0, K), (256, min(768, K)), (512, K)):
            if k_begin >= k_end:
                continue
            folds = {k for k in range(k_begin + 256, k_end, 256)}
            assert all(k % 2 == 0 for k in folds)
            for kt in range(k_begin // 18, -(-k_end // 18)):
                kbase = kt * 18
                interior = kbase >= k_begin and kbase + 18 <= k_end and ((kbase + 17) >> 8) == (kbase >> 8) and ((kbase & 255) != 0 or kbase == k_begin)
                needs_checks = kbase < k_begin or kbase + 18 > k_end or any(kbase <= f < kbase + 18 for f in folds)
                assert interior == (not needs_checks), (K, k_begin, k_end, kt)
