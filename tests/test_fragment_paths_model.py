"""k_fine merges the fragments of a batch of 64 coverage masks into per-pixel keys `depth << 32 | ~id` by three routes
(raster.hip raster_pairs): masks of up to eight fragments are popped bit by bit by their own lanes; larger ones are taken by the
whole wave, lane = pixel, or -- five or more of them in a batch -- eight at a time, eight lanes per mask, lane k of a group
walking column k of its mask's tile downwards.  The routes differ in WHO computes a fragment, never in what it is: pixel
(b & 7, b >> 3) of mask bit b, depth d0 + zx * x + zy * y in U32 wrap-around arithmetic (FineRaster.inl:348), key minimum per pixel.
This file restates the three index / depth computations in numpy, lane by lane as the kernel does them (the 24-bit multiply
pieces, the incremental adds, the shifted mask halves, the idle groups), and checks that every batch gives the same key array
by all of them.  The GPU-side proof is the bit-exact parity of tests/test_gpu_fuzz.py and tools/fuzz_soak.py; this is the
arithmetic on paper."""
import numpy as np
import pytest

U32 = 0xFFFFFFFF
KINIT = (0xFFFFFFFF << 32) | 0xFFFFFFFF


def _umul24(a, b):
    return ((a & 0xFFFFFF) * (b & 0xFFFFFF)) & U32


def _depth_pieces(d0, zx, zy, x, y):
    """The per-lane and wave-wide routes: zx * x + zy * y through 24-bit multiplies of the low 24 and the high 8 bits."""
    return (d0 + _umul24(zx & 0xFFFFFF, x) + ((_umul24(zx >> 24, x) << 24) & U32)
            + _umul24(zy & 0xFFFFFF, y) + ((_umul24(zy >> 24, y) << 24) & U32)) & U32


def _merge(keys, tile, pix, depth, idk):
    k = (depth << 32) | idk
    if k < keys[tile][pix]:
        keys[tile][pix] = k


def _route_pop(masks, planes, tiles, keys, pick):
    for lane in pick:
        m = masks[lane]
        d0, zx, zy, idk = planes[lane]
        while m:
            b = (m & -m).bit_length() - 1
            m &= m - 1
            _merge(keys, tiles[lane], b, _depth_pieces(d0, zx, zy, b & 7, b >> 3), idk)


def _route_wave(masks, planes, tiles, keys, pick):
    for src in pick:                                         # one pass per mask; lane = pixel, the mask is the execution mask
        d0, zx, zy, idk = planes[src]
        for lane in range(64):
            if (masks[src] >> lane) & 1:
                _merge(keys, tiles[src], lane, _depth_pieces(d0, zx, zy, lane & 7, lane >> 3), idk)


def _route_octets(masks, planes, tiles, keys, pick):
    heavy = list(pick)
    while heavy:
        group, heavy = heavy[:8], heavy[8:]
        for lane in range(64):
            g, k = lane >> 3, lane & 7
            src = group[g] if g < len(group) else -1         # fewer than eight left: idle groups
            if src < 0:
                continue
            smlo, smhi = masks[src] & U32, masks[src] >> 32
            d0, zx, zy, idk = planes[src]
            wlo, whi = smlo >> k, smhi >> k                  # column k: bit 8 * (y & 3) of the half that holds row y
            depth = (d0 + zx * k) & U32
            for y in range(8):
                if ((wlo if y < 4 else whi) >> (8 * (y & 3))) & 1:
                    _merge(keys, tiles[src], y * 8 + k, depth, idk)
                depth = (depth + zy) & U32


def _batch(rng, kind):
    masks, planes, tiles = [], [], []
    for lane in range(64):
        if kind == "full":
            m = U32 << 32 | U32
        elif kind == "sparse":
            m = int(rng.integers(0, 1 << 62)) & int(rng.integers(0, 1 << 62)) & int(rng.integers(0, 1 << 62))
        else:
            m = int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63)
            if rng.uniform() < 0.3:
                m &= (0xFF << (8 * int(rng.integers(0, 8)))) | (0x0101010101010101 << int(rng.integers(0, 8)))
            if rng.uniform() < 0.1:
                m = 0
        masks.append(m)
        # depth planes with wrap-around slopes (negative slopes are large U32 values) and ids in the key's low word
        planes.append((int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))))
        tiles.append(int(rng.integers(0, 4)))                 # few tiles: many masks meet in one key array
    return masks, planes, tiles


@pytest.mark.parametrize("kind", ["random", "sparse", "full"])
def test_the_three_fragment_routes_build_the_same_keys(kind):
    rng = np.random.default_rng({"random": 1, "sparse": 2, "full": 3}[kind])
    for _ in range(40):
        masks, planes, tiles = _batch(rng, kind)
        every = range(64)
        ref = [[KINIT] * 64 for _ in range(4)]
        _route_pop(masks, planes, tiles, ref, every)
        for route in (_route_wave, _route_octets):
            keys = [[KINIT] * 64 for _ in range(4)]
            route(masks, planes, tiles, keys, every)
            assert keys == ref, route.__name__
        # ... and as the kernel splits a batch: small masks popped, big ones in octets while five or more remain, then wave-wide
        big = [l for l in every if bin(masks[l]).count("1") > 8]
        small = [l for l in every if l not in big]
        keys = [[KINIT] * 64 for _ in range(4)]
        _route_pop(masks, planes, tiles, keys, small)
        rest = list(big)
        while len(rest) >= 5:
            _route_octets(masks, planes, tiles, keys, rest[:8])
            rest = rest[8:]
        _route_wave(masks, planes, tiles, keys, rest)
        assert keys == ref


def test_incremental_depth_equals_the_24_bit_pieces():
    """The octet route multiplies zx by the column once and adds zy per row; the other routes assemble zx * x + zy * y from
    24-bit multiplies.  Both are the product modulo 2^32."""
    rng = np.random.default_rng(4)
    for _ in range(2000):
        d0, zx, zy = (int(v) for v in rng.integers(0, 1 << 32, size=3))
        for x in range(8):
            d = (d0 + zx * x) & U32
            for y in range(8):
                assert d == _depth_pieces(d0, zx, zy, x, y) == (d0 + zx * x + zy * y) & U32
                d = (d + zy) & U32
