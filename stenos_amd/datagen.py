"""Seeded synthetic typed-array generators shared by tests, golden fixtures and bench.py.

All integer generators are built on splitmix64 (one draw per element), so that numpy on the
host and torch on the device produce the same bytes (SURVEY.md section 8d proposes this generator;
the reference itself only has std::mt19937-based test inputs, tests/tests_comp_decomp.cpp:37-86).

    s += 0x9E3779B97F4A7C15; z = s
    z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9
    z = (z ^ z >> 27) * 0x94D049BB133111EB
    u = z ^ z >> 31                                   (all mod 2**64)
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, start: int = 0) -> np.ndarray:
    """u[i] for i in [start, start+n): the (i+1)-th output of splitmix64 seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _as_bytes(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


def _le_elements(x: np.ndarray, T: int) -> np.ndarray:
    """int64 values -> n elements of T little-endian bytes (sign-extended above 8 bytes)."""
    n = x.shape[0]
    b = np.ascontiguousarray(x.astype("<i8")).view(np.uint8).reshape(n, 8)
    out = np.empty((n, T), dtype=np.uint8)
    k = min(T, 8)
    out[:, :k] = b[:, :k]
    if T > 8:
        out[:, 8:] = np.where(x[:, None] < 0, 255, 0).astype(np.uint8)
    return out.reshape(-1)


def generate(kind: str, T: int, n: int, seed: int = 42) -> np.ndarray:
    """Return n elements of T bytes as a flat uint8 array.

    kinds: sorted_i32 (README example, T=4), rand (full entropy), rand12 (u & 0xFFF, T=4), rand8 (u & 0xFF),
    same, sorted (lexicographically sorted random elements, signed bytes as std::array<char,N>),
    walk (x += u%17 - 8), dict16 (16-entry dictionary), runs (runs of 7), burst, ramp,
    sine (float64 sin(i*0.001), T=8; float32 for T=4), smooth8 (config 5b byte signal),
    mixed (per block and plane: constant / narrow / random / walk / runs), lzmix (per block: noise / dictionary /
    half dictionary / constant, 12-bit values), noise_low (noise below a slow ramp), steps (long runs of equal values),
    slopes (piecewise linear), cycle130 (a cycle of 130 values), edge_noise (planes of noise with a few rows squeezed,
    repeated or smoothed: plane sizes on both sides of the 256 bytes above which a plane is stored raw).
    """
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    u = splitmix64(seed, n)
    if kind == "sorted_i32":
        assert T == 4
        return _as_bytes(np.arange(n, dtype="<i4"))
    if kind == "rand":
        m = splitmix64(seed, (n * T + 7) // 8)
        return _as_bytes(m)[: n * T].copy()
    if kind == "rand12":
        assert T == 4
        return _as_bytes((u & np.uint64(0xFFF)).astype("<u4"))
    if kind == "rand8":  # small integers in wide elements: one non-constant plane
        return _le_elements((u & np.uint64(0xFF)).astype(np.int64), T)
    if kind == "same":
        return np.full(n * T, int(u[0] & np.uint64(0xFF)), dtype=np.uint8)
    if kind == "sorted":
        a = generate("rand", T, n, seed).reshape(n, T)
        order = np.lexsort(a.view(np.int8).T[::-1])
        return np.ascontiguousarray(a[order]).reshape(-1)
    if kind == "walk":
        steps = (u % np.uint64(17)).astype(np.int64) - 8
        return _le_elements(np.cumsum(steps), T)
    if kind == "ramp":
        return _le_elements(np.arange(n, dtype=np.int64) * 3 + 1000, T)
    if kind == "dict16":
        d = generate("rand", T, 16, seed + 1).reshape(16, T)
        return np.ascontiguousarray(d[(u & np.uint64(15)).astype(np.int64)]).reshape(-1)
    if kind == "runs":
        v = generate("rand", T, n // 7 + 1, seed + 1).reshape(-1, T)
        return np.ascontiguousarray(np.repeat(v, 7, axis=0)[:n]).reshape(-1)
    if kind == "burst":
        # mostly-constant elements with short random bursts and a slow step pattern
        a = np.zeros((n, T), dtype=np.uint8)
        hot = (u % np.uint64(50)) == 0
        burst = np.convolve(hot.astype(np.int32), np.ones(9, dtype=np.int32))[:n] > 0
        r = generate("rand", T, n, seed + 2).reshape(n, T)
        a[burst] = r[burst] & 0x3F
        a += ((np.arange(n) // 64) % 3).astype(np.uint8)[:, None]
        return a.reshape(-1)
    if kind == "mixed":
        # every plane of every 256-element block has a style of its own (constant, narrow, random, walk, runs):
        # neighbouring blocks differ in how many planes are constant
        nb = (n + 255) // 256
        style = (splitmix64(seed + 7, nb * T) % np.uint64(6)).astype(np.int64).reshape(nb, 1, T)
        r = generate("rand", T, nb * 256, seed + 3).reshape(nb, 256, T)
        const = np.broadcast_to(r[:, :1, :], r.shape)
        narrow = (r & 0x0F) + const
        walk = np.cumsum((r % 5).astype(np.int64) - 2, axis=1).astype(np.uint8) + const
        runs = np.repeat(r[:, ::16, :], 16, axis=1)
        a = np.where(style <= 1, const, np.where(style == 2, narrow, np.where(style == 3, r, np.where(style == 4, walk, runs))))
        return np.ascontiguousarray(a.astype(np.uint8).reshape(nb * 256, T)[:n]).reshape(-1)
    if kind == "lzmix":
        # per 256-element block one of: 12-bit noise (no LZ), a 16-entry dictionary of 12-bit values (mini-LZ succeeds),
        # a dictionary in the first half and noise behind it (an attempt that starts and fails), a constant block
        nb = (n + 255) // 256
        style = (splitmix64(seed + 11, nb) % np.uint64(4)).astype(np.int64)[:, None]
        r = (splitmix64(seed + 5, nb * 256) & np.uint64(0xFFF)).astype(np.int64).reshape(nb, 256)
        d = (splitmix64(seed + 6, nb * 16) & np.uint64(0xFFF)).astype(np.int64).reshape(nb, 16)
        pick = np.take_along_axis(d, (r & 15), axis=1)
        half = np.where(np.arange(256)[None, :] < 128, pick, r)
        a = np.where(style == 0, r, np.where(style == 1, pick, np.where(style == 2, half, d[:, :1])))
        return _le_elements(a.reshape(-1)[:n], T)
    if kind == "sine":
        x = np.sin(np.arange(n, dtype=np.float64) * 0.001)
        if T == 8:
            return _as_bytes(x.astype("<f8"))
        if T == 4:
            return _as_bytes(x.astype("<f4"))
        raise ValueError("sine needs T in (4, 8)")
    if kind == "noise_low":
        # noise in the low half of the element, a slow ramp above it (the byte planes of doubles: whole passes of RAW planes)
        h = max(1, T // 2)
        a = generate("rand", T, n, seed + 13).reshape(n, T).copy()
        a[:, h:] = _le_elements((np.arange(n, dtype=np.int64) // 3 + seed), T).reshape(n, T)[:, : T - h]
        return a.reshape(-1)
    if kind == "steps":
        # long runs of equal values with jumps: most rows are run-length rows of values (row header 7)
        v = (splitmix64(seed + 17, n // 9 + 2) & np.uint64(0x3FFFFFFF)).astype(np.int64)
        return _le_elements(np.repeat(v, 9)[:n], T)
    if kind == "slopes":
        # piecewise linear: differences that repeat -- run-length rows of differences (row header 6)
        d = (splitmix64(seed + 19, n // 11 + 2) % np.uint64(200)).astype(np.int64)
        return _le_elements(np.cumsum(np.repeat(d, 11)[:n]), T)
    if kind == "cycle130":
        # 130 distinct values over and over: the mini-LZ finds every value 130 items back (distances of two bytes)
        v = (splitmix64(seed + 23, 130) >> np.uint64(2)).astype(np.int64)  # (noise in every byte: the attempt is not turned away early)
        return _le_elements(np.tile(v, n // 130 + 1)[:n], T)
    if kind == "edge_noise":
        # Planes of noise in which some rows of 16 bytes are made a little compressible: squeezed into a span of 60..68,
        # given two to four repeated bytes or repeated differences, or made smooth.  A plane's size then lands within a few
        # bytes of 256, on either side: the RAW decision (block_compress.h:1200-1204) and every shortcut around it are tested
        # where they could go wrong.  One plane in four stays pure noise, one in eight is constant.
        nb = (n + 255) // 256
        rng = np.random.default_rng(seed * 7919 + T)
        a = rng.integers(0, 256, size=(nb, T, 16, 16), dtype=np.int64)
        plane_style = rng.integers(0, 8, size=(nb, T))
        nrows = rng.integers(0, 5, size=(nb, T))  # rows of the plane that are touched
        for b in range(nb):
            for t in range(T):
                if plane_style[b, t] == 0:
                    a[b, t] = a[b, t, 0, 0]
                    continue
                if plane_style[b, t] <= 2:
                    continue
                for r in rng.choice(16, size=int(nrows[b, t]), replace=False):
                    mode = int(rng.integers(0, 4))
                    row = a[b, t, r]
                    if mode == 0:
                        w = int(rng.integers(60, 69))
                        c = int(rng.integers(0, 256))
                        row[:] = (c + row % w) & 0xFF
                    elif mode == 1:
                        for c in rng.choice(np.arange(1, 16), size=int(rng.integers(2, 5)), replace=False):
                            row[c] = row[c - 1]
                    elif mode == 2:
                        for c in sorted(rng.choice(np.arange(2, 16), size=int(rng.integers(2, 5)), replace=False)):
                            row[c] = (2 * row[c - 1] - row[c - 2]) & 0xFF
                    else:
                        row[:] = (int(row[0]) + np.arange(16) * int(rng.integers(0, 3)) + row % int(rng.integers(1, 9))) & 0xFF
        e = a.reshape(nb, T, 256).transpose(0, 2, 1).astype(np.uint8)  # plane t of block b -> byte t of its 256 elements
        return np.ascontiguousarray(e.reshape(nb * 256, T)[:n]).reshape(-1)
    if kind == "smooth8":
        assert T == 1
        x = (128 + 100 * np.sin(0.01 * np.arange(n))).astype(np.int64) + (u % np.uint64(5)).astype(np.int64) - 2
        return (x & 0xFF).astype(np.uint8)
    raise ValueError(f"unknown kind {kind}")


# ------------------------------------------------------------------------------------------------
# device-side generation (torch), same sequences as the numpy generators above
# ------------------------------------------------------------------------------------------------
def _lsr(z, k: int):
    """logical shift right of an int64 torch tensor"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_torch(seed: int, n: int, device, start: int = 0):
    import torch

    def wrap(v: int) -> int:  # two's complement int64 view of a uint64 constant
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    idx = torch.arange(start + 1, start + n + 1, dtype=torch.int64, device=device)
    z = idx * wrap(0x9E3779B97F4A7C15) + wrap(seed)
    z = (z ^ _lsr(z, 30)) * wrap(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * wrap(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def generate_torch(kind: str, T: int, n: int, seed: int = 42, device="cuda", chunk: int = 1 << 25, start: int = 0):
    """Flat uint8 CUDA tensor holding elements [start, start + n) of the sequence `kind` (T bytes each);
    kinds: sorted_i32, rand, rand12, rand8, walk (start must be 0), sine, dict16, steps, smooth8."""
    import torch

    out = torch.empty(n * T, dtype=torch.uint8, device=device)
    carry = 0
    assert start == 0 or kind != "walk"
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        s = s0 + start
        if kind == "sorted_i32":
            v = torch.arange(s, s + m, dtype=torch.int64, device=device).to(torch.int32)
        elif kind == "rand":
            assert (T * chunk) % 8 == 0
            nb = (m * T + 7) // 8
            v = splitmix64_torch(seed, nb, device, start=s * T // 8).view(torch.uint8)[: m * T]
        elif kind == "rand12":
            v = (splitmix64_torch(seed, m, device, start=s) & 0xFFF).to(torch.int32)
        elif kind == "rand8":
            v = (splitmix64_torch(seed, m, device, start=s) & 0xFF).to({2: torch.int16, 4: torch.int32, 8: torch.int64}[T])
        elif kind == "walk":
            u = splitmix64_torch(seed, m, device, start=s)
            # u mod 17 for the unsigned 64-bit value: (hi * 2^32 + lo) mod 17 with 2^32 mod 17 = 1
            lo, hi = u & 0xFFFFFFFF, _lsr(u, 32)
            steps = (hi + lo) % 17 - 8
            x = torch.cumsum(steps, 0) + carry
            carry = int(x[-1].item())
            v = {2: torch.int16, 4: torch.int32, 8: torch.int64}[T]
            v = x.to(v)
        elif kind == "sine":
            x = torch.sin(torch.arange(s, s + m, dtype=torch.float64, device=device) * 0.001)
            v = x if T == 8 else x.to(torch.float32)
        elif kind == "dict16":
            d = torch.from_numpy(generate("rand", T, 16, seed + 1).reshape(16, T)).to(device)
            v = d[(splitmix64_torch(seed, m, device, start=s) & 15)]
        elif kind == "steps":  # (generate("steps"): element i = value number i // 9)
            idx = torch.arange(s, s + m, dtype=torch.int64, device=device) // 9
            z = (idx + 1) * (0x9E3779B97F4A7C15 - (1 << 64)) + (seed + 17)
            z = (z ^ _lsr(z, 30)) * (0xBF58476D1CE4E5B9 - (1 << 64))
            z = (z ^ _lsr(z, 27)) * (0x94D049BB133111EB - (1 << 64))
            v = ((z ^ _lsr(z, 31)) & 0x3FFFFFFF).to({2: torch.int16, 4: torch.int32, 8: torch.int64}[T])
        elif kind == "smooth8":  # (the device's sin: the same signal as generate(), not bit for bit the same bytes)
            assert T == 1
            u = splitmix64_torch(seed, m, device, start=s)
            lo, hi = u & 0xFFFFFFFF, _lsr(u, 32)
            noise = (hi * 1 + lo) % 5 - 2  # u mod 5 of the unsigned value: 2^32 mod 5 = 1
            x = (128 + 100 * torch.sin(0.01 * torch.arange(s, s + m, dtype=torch.float64, device=device))).to(torch.int64) + noise
            v = (x & 0xFF).to(torch.uint8)
        else:
            raise ValueError(kind)
        out[s0 * T:(s0 + m) * T] = v.contiguous().view(torch.uint8).reshape(-1)
    return out
