"""splitmix64 stream (all synthetic inputs come from splitmix64)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


class SplitMix64:
    def __init__(self, seed):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = self.state + np.uint64(n) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def randint(self, lo, hi, size=None):
        """uniform integers in [lo, hi] (inclusive)"""
        n = int(np.prod(size)) if size is not None else 1
        r = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        v = (lo + np.floor(r * (hi - lo + 1))).astype(np.int64)
        return v.reshape(size) if size is not None else int(v[0])

    def u8(self, size):
        return self.randint(0, 255, size).astype(np.uint8)

    def uniform(self, size=None):
        n = int(np.prod(size)) if size is not None else 1
        r = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        return r.reshape(size) if size is not None else float(r[0])

    def laplace_int(self, b, size, lim):
        u = self.uniform(size) - 0.5
        v = -b * np.sign(u) * np.log1p(-2.0 * np.abs(u) + 1e-300)
        return np.clip(np.rint(v), -lim, lim).astype(np.int64)
