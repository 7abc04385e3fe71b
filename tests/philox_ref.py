"""Pure-Python Philox4x32-10 + Box-Muller, the published algorithm (Salmon et al. 2011) the
device generator implements; used only to pin the device normals in the GPU tests."""
import math

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c, k0, k1):
    c = list(c)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c


def philox_normal(seed, dof, sample):
    c = philox4x32_10([dof & MASK, dof >> 32, sample & MASK, sample >> 32], seed & MASK, seed >> 32)
    a = (c[1] << 32) | c[0]
    b = (c[3] << 32) | c[2]
    u1 = ((a >> 11) + 0.5) / 9007199254740992.0
    u2 = ((b >> 11) + 0.5) / 9007199254740992.0
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586476925 * u2)


def philox_normals_np(seed, n, first_id, k):
    """Vectorised NumPy version: (n, k) array, column s = sample id first_id + s."""
    import numpy as np
    dof = np.arange(n, dtype=np.uint64)[:, None] * np.ones((1, k), dtype=np.uint64)
    sid = (np.uint64(first_id) + np.arange(k, dtype=np.uint64))[None, :] * np.ones((n, 1), dtype=np.uint64)
    m32 = np.uint64(MASK)
    c = [dof & m32, dof >> np.uint64(32), sid & m32, sid >> np.uint64(32)]
    k0, k1 = np.uint64(seed & MASK), np.uint64(seed >> 32)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + np.uint64(W0)) & m32
        k1 = (k1 + np.uint64(W1)) & m32
    a = (c[1] << np.uint64(32)) | c[0]
    b = (c[3] << np.uint64(32)) | c[2]
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)
