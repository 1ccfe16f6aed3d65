"""Host-side model of the fragment-tiled layouts and of the MFMA lane maps the kernels rely on.

A numpy emulation of v_mfma_f32_32x32x16_bf16 (operand/accumulator lane maps as documented for
gfx950) is used to check, without a GPU, that (a) the fragment-tiled index gives lane-linear operand
fragments, (b) the "swapped" epilogue index math lands on the right (token, feature), and (c) the
key permutation written by the V epilogue is the one the S^T accumulator presents to the PV MFMA.
"""
import numpy as np


def tiled_index(r, k, K16):
    return ((((r >> 5) * K16 + (k >> 4)) * 2 + ((k >> 3) & 1)) << 8) + ((r & 31) << 3) + (k & 7)


def frag_from_tiled(plane, rt, ks, K16):
    """What a wave reads: lane l takes 8 elements at block_base + 8*l."""
    base = ((rt * K16 + ks) * 2) << 8
    return plane[base:base + 512].reshape(64, 8)


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag/b_frag: [64 lanes][8]; acc: [64 lanes][16].  A[i][k]: lane = i + 32*(k//8), elem k%8;
    B[k][j]: lane = j + 32*(k//8); D[i][j]: lane = j + 32*((i%8)//4), reg = (i%4) + 4*(i//8)."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        for e in range(8):
            A[l & 31, 8 * (l >> 5) + e] = a_frag[l, e]
            Bm[8 * (l >> 5) + e, l & 31] = b_frag[l, e]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def test_tiled_index_is_a_bijection_and_lane_linear():
    R, K = 64, 48
    idx = np.array([[tiled_index(r, k, K // 16) for k in range(K)] for r in range(R)])
    assert sorted(idx.ravel().tolist()) == list(range(R * K))
    plane = np.zeros(R * K)
    M = np.arange(R * K, dtype=np.float64).reshape(R, K)
    plane[idx.ravel()] = M.ravel()
    f = frag_from_tiled(plane, 1, 2, K // 16)
    for l in range(64):
        assert np.array_equal(f[l], M[32 + (l & 31), 32 + 8 * (l >> 5): 32 + 8 * (l >> 5) + 8])


def test_swapped_gemm_epilogue_addresses():
    rng = np.random.default_rng(0)
    N, M, K = 32, 32, 32
    W, X = rng.standard_normal((N, K)), rng.standard_normal((M, K))
    wp, xp = np.zeros(N * K), np.zeros(M * K)
    for r in range(32):
        for k in range(K):
            wp[tiled_index(r, k, K // 16)] = W[r, k]
            xp[tiled_index(r, k, K // 16)] = X[r, k]
    acc = np.zeros((64, 16))
    for ks in range(K // 16):
        acc = mfma_32x32x16(frag_from_tiled(wp, 0, ks, K // 16), frag_from_tiled(xp, 0, ks, K // 16), acc)
    # epilogue of gemm.h: lane owns token m = lane & 31; reg r is feature 8*(r>>2) + 4*hf + (r&3)
    out = np.zeros(M * N)
    for l in range(64):
        hf, col = l >> 5, l & 31
        for g in range(4):
            f = 8 * g + 4 * hf
            for c in range(4):
                out[tiled_index(col, f, N // 16) + c] = acc[l, 4 * g + c]
    Y = X @ W.T
    for m in range(M):
        for f in range(N):
            assert abs(out[tiled_index(m, f, N // 16)] - Y[m, f]) < 1e-9


def swap23(k):
    return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1)


def test_accumulator_order_chain_of_two_gemms():
    """Epilogues store activations in 'accumulator order' (feature bits 2 and 3 swapped inside each group of
    16) with ONE 16-byte slot per lane per (tile, jj); the consumer's weights are packed with the same
    permutation along K.  Chain two emulated GEMMs and compare with the plain matmul."""
    assert all(swap23(swap23(k)) == k for k in range(64))
    rng = np.random.default_rng(3)
    N1, N2, M, K = 32, 32, 32, 32
    W1, W2, X = rng.standard_normal((N1, K)), rng.standard_normal((N2, N1)), rng.standard_normal((M, K))
    w1p, xp = np.zeros(N1 * K), np.zeros(M * K)
    for r in range(32):
        for k in range(K):
            w1p[tiled_index(r, k, K // 16)] = W1[r, k]       # first GEMM consumes a natural-order operand
            xp[tiled_index(r, k, K // 16)] = X[r, k]
    acc = np.zeros((64, 16))
    for ks in range(K // 16):
        acc = mfma_32x32x16(frag_from_tiled(w1p, 0, ks, K // 16), frag_from_tiled(xp, 0, ks, K // 16), acc)
    # epilogue (gemm.h acc_slot): lane (col, hf), jj: regs 8jj..8jj+7 -> 8 contiguous elements of block ks16 = jj, half hf
    yp = np.zeros(M * N1)
    for l in range(64):
        hf, col = l >> 5, l & 31
        for jj in range(2):
            base = ((0 * (N1 // 16) + jj) * 2 + hf) * 256 + col * 8
            yp[base:base + 8] = acc[l, 8 * jj:8 * jj + 8]
    Y = X @ W1.T
    for m in range(M):
        for f in range(N1):
            assert abs(yp[tiled_index(m, swap23(f), N1 // 16)] - Y[m, f]) < 1e-9
    # second GEMM: weights packed in accumulator order along K
    w2p = np.zeros(N2 * N1)
    for r in range(N2):
        for k in range(N1):
            w2p[tiled_index(r, swap23(k), N1 // 16)] = W2[r, k]
    acc2 = np.zeros((64, 16))
    for ks in range(N1 // 16):
        acc2 = mfma_32x32x16(frag_from_tiled(w2p, 0, ks, N1 // 16), frag_from_tiled(yp, 0, ks, N1 // 16), acc2)
    Z = Y @ W2.T
    for l in range(64):
        for r in range(16):
            assert abs(acc2[l, r] - Z[l & 31, (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]) < 1e-9


def test_v_key_permutation_matches_softmax_accumulator():
    """S^T accumulator regs 8jj..8jj+7 of a lane, used verbatim as the PV B-fragment, pair with the V^T
    fragment written by EpiV: slot (hf, e) <-> key 16*kg + 8*(e>>2) + 4*hf + (e&3)."""
    rng = np.random.default_rng(1)
    Lk, D = 32, 32
    P = rng.standard_normal((32, Lk))          # P[query][key]
    V = rng.standard_normal((Lk, D))           # V[key][d]
    # S^T accumulator layout: lane (q = l&31, hf), reg r <-> key (r&3) + 8*(r>>2) + 4*hf
    sacc = np.zeros((64, 16))
    for l in range(64):
        for r in range(16):
            sacc[l, r] = P[l & 31, (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]
    # EpiV: [dt][kg][hf][d%32][e], key%16 = 8a + 4b + c -> slot (hfk=b, e=4a+c)
    vt = np.zeros((Lk // 16, 2, 32, 8))
    for key in range(Lk):
        k16 = key & 15
        a, b, c = k16 >> 3, (k16 >> 2) & 1, k16 & 3
        vt[key >> 4, b, :, 4 * a + c] = V[key, :]
    acc = np.zeros((64, 16))
    for kg in range(Lk // 16):
        jj = kg & 1
        pfrag = sacc[:, 8 * jj:8 * jj + 8] if (kg >> 1) == 0 else None
        vfrag = vt[kg].reshape(64, 8)
        acc = mfma_32x32x16(vfrag, pfrag, acc)
    O = P @ V   # [query][d]
    for l in range(64):
        for r in range(16):
            d = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
            assert abs(acc[l, r] - O[l & 31, d]) < 1e-9


def test_xcd_remap_is_bijective():
    def remap(bid, n):
        q, r = n >> 3, n & 7
        xcd, idx = bid & 7, bid >> 3
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
    for n in (1, 7, 8, 9, 24, 100, 6144, 6151):
        assert sorted(remap(b, n) for b in range(n)) == list(range(n))


# ------------------------------------------------------------------ int8 slices ("i8x3", csrc/common.h, gemm.h)
QMAX = 32639


def acc32(f):
    r = f & 31
    return (f & ~31) | (((r >> 2) & 1) << 4) | ((r >> 3) << 2) | (r & 3)


def tiled_index_i8(r, k, K32):
    return ((((r >> 5) * K32 + (k >> 5)) * 2 + ((k >> 4) & 1)) << 9) + ((r & 31) << 4) + (k & 15)


def slices(v, scale_inv):
    """quant16: q = rint(v * inv) through the float adder (magic number), low byte = s2, byte 1 of q + 128 = s1."""
    u = (np.float32(v) * np.float32(scale_inv) + np.float32(12582912.0)).astype(np.float32).view(np.uint32)
    s2 = (u & 0xFF).astype(np.uint8).view(np.int8)
    s1 = (((u + np.uint32(128)) >> 8) & 0xFF).astype(np.uint8).view(np.int8)
    return s1.astype(np.int64), s2.astype(np.int64)


def test_int8_slices_reconstruct_the_16_bit_integer():
    rng = np.random.default_rng(0)
    v = rng.standard_normal(4096).astype(np.float32)
    v[:4] = [0.0, v.max(), -np.abs(v).max(), np.abs(v).max()]
    inv = np.float32(QMAX) / np.abs(v).max()
    s1, s2 = slices(v, inv)
    q = np.rint(v.astype(np.float64) * np.float64(inv))
    assert np.abs(256 * s1 + s2 - q).max() <= 1  # fma rounds once, rint(float product) twice: at most one unit apart
    assert np.abs(s1).max() <= 127 and s2.min() >= -128 and s2.max() <= 127
    # the two-accumulator product: 256*H + M fits int32 for K <= 512 and reproduces the dropped-s2*s2 product
    K = 512
    worst = K * (127 * 127 * 256 + 2 * 127 * 128)
    assert worst < 2 ** 31
    a1, a2 = slices(rng.standard_normal(K).astype(np.float32), np.float32(9000.0))
    w1, w2 = slices(rng.standard_normal(K).astype(np.float32), np.float32(9000.0))
    H, M = int((a1 * w1).sum()), int((a1 * w2 + a2 * w1).sum())
    exact = int(((256 * a1 + a2) * (256 * w1 + w2)).sum())
    assert 256 * (256 * H + M) == exact - int((a2 * w2).sum())


def test_acc32_order_and_int8_fragment_layout():
    # acc32 permutes inside groups of 32 and sends accumulator register 4g + c of lane half hf (feature 8g + 4hf + c) to byte 16hf + 4g + c
    assert sorted(acc32(f) for f in range(64)) == list(range(64))
    for hf in range(2):
        for g in range(4):
            for c in range(4):
                assert acc32(32 + 8 * g + 4 * hf + c) == 32 + 16 * hf + 4 * g + c
    # tiled_index_i8: bijection; a lane (row, half) of an MFMA fragment owns 16 consecutive bytes at 16 * lane
    R, K = 64, 128
    idx = np.array([[tiled_index_i8(r, k, K // 32) for k in range(K)] for r in range(R)])
    assert sorted(idx.ravel()) == list(range(R * K))
    for r in (0, 5, 37):
        for kb in (0, 3):
            for half in range(2):
                base = idx[r, kb * 32 + half * 16]
                assert base % 1024 == 16 * (half * 32 + (r & 31))
                assert np.array_equal(idx[r, kb * 32 + half * 16: kb * 32 + half * 16 + 16], base + np.arange(16))
    # a contraction is indifferent to the K order as long as both operands share it
    rng = np.random.default_rng(1)
    a, w = rng.integers(-127, 128, (4, 64)), rng.integers(-127, 128, (3, 64))
    perm = np.array([acc32(k) for k in range(64)])
    ap, wp = np.empty_like(a), np.empty_like(w)
    ap[:, perm], wp[:, perm] = a, w
    assert np.array_equal(a @ w.T, ap @ wp.T)

