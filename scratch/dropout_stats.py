"""Statistics of the counter-based dropout mask generators (numpy emulation of csrc/common.h: dropout_mask8): keep rate per position,
pairwise correlation of the 8 keep decisions of a block, the 256-pattern chi-square against the binomial law, lag correlations between
consecutive blocks and between two layers (different offsets).  usage: python scratch/dropout_stats.py [n_blocks]"""
import sys
import numpy as np
M = np.uint32(0xffffffff)


def lowbias32(x):
	x = x.astype(np.uint32)
	x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)) & M
	x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b)) & M
	x ^= x >> np.uint32(16)
	return x


def xorshift32(x):
	x = x.copy()
	x ^= (x << np.uint32(13)) & M; x ^= x >> np.uint32(17); x ^= (x << np.uint32(5)) & M
	return x


def key_of(seed, c4_hi):
	return np.uint32(seed & 0xffffffff) ^ lowbias32(np.array([c4_hi ^ (seed >> 32)], dtype = np.uint32))[0]


def words_hash4(lo, key):  # the round-2 generator: four hashed words per block
	return [lowbias32((lo + np.uint32(i)) ^ key) for i in range(4)]


def words_hash1_xs3(lo, key):  # candidate: one hashed word, three xorshift32 steps
	r0 = lowbias32(lo ^ key)
	r1 = xorshift32(r0); r2 = xorshift32(r1); r3 = xorshift32(r2)
	return [r0, r1, r2, r3]


def keep_bits(words, thr):
	cols = []
	for r in words:
		cols.append((r & np.uint32(0xffff)) >= thr)
		cols.append((r >> np.uint32(16)) >= thr)
	return np.stack(cols, 1)  # (n, 8) bool


def report(name, gen, n, p = 0.2, seed = 0x5EEDC0DE12345678):
	thr = np.uint32(round(p * 65536))
	q = 1.0 - float(thr) / 65536
	lo = (np.arange(n, dtype = np.uint64) * 4 + 4 * 1000003).astype(np.uint32)
	k = keep_bits(gen(lo, key_of(seed, 0)), thr).astype(np.float64)
	rate = k.mean(0)
	c = np.corrcoef(k.T)
	off = np.abs(c - np.eye(8)).max()
	pat = (k.astype(np.int64) * (1 << np.arange(8))).sum(1)
	cnt = np.bincount(pat, minlength = 256).astype(np.float64)
	ones = np.array([bin(i).count('1') for i in range(256)])
	exp = n * q ** ones * (1 - q) ** (8 - ones)
	chi2 = ((cnt - exp) ** 2 / exp).sum()
	lag1 = max(abs(np.corrcoef(k[:-1, i], k[1:, j])[0, 1]) for i in range(8) for j in range(8))
	k2 = keep_bits(gen(lo + np.uint32(4 * 12345677), key_of(seed, 0)), thr).astype(np.float64)  # another layer: offset + 12345677 blocks
	layer = max(abs(np.corrcoef(k[:, i], k2[:, j])[0, 1]) for i in range(8) for j in range(8))
	k3 = keep_bits(gen(lo, key_of(seed + 1, 0)), thr).astype(np.float64)  # another seed
	seedc = max(abs(np.corrcoef(k[:, i], k3[:, j])[0, 1]) for i in range(8) for j in range(8))
	sig = 1 / np.sqrt(n)
	print(f'{name}: n = {n} blocks; keep rate {rate.min():.5f}..{rate.max():.5f} (expected {q:.5f}, 3 sigma {3 * np.sqrt(q * (1 - q) / n):.5f}); max |corr| within a block {off:.2e}, lag 1 {lag1:.2e}, other layer {layer:.2e}, other seed {seedc:.2e} (3 sigma {3 * sig:.2e}); chi2 of the 256 patterns {chi2:.1f} (255 dof: 95 % < 293, 99.9 % < 331)')


if __name__ == '__main__':
	n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
	report('hash4 (round 2)', words_hash4, n)
	report('hash1 + 3 xorshift32', words_hash1_xs3, n)
