"""k_fine's coverage walk evaluates two pixels per instruction in the 16-bit halves of a register (raster.hip raster_pairs,
NVDR_FINE_PK16).  This is the arithmetic of that walk restated in numpy -- reduction by four bits, start values clamped to
+-16000, rows 0..3 in the low half and 4..7 in the high half, 16-bit wrap-around -- against the integer fill rule of
SURVEY App. A3 (`Util.inl:304-309`) on random and adversarial edge functions: every 8x8 mask must be identical and no
intermediate may leave the 16-bit range for edges the kernel sends down this path (|A/16| + |B/16| <= 2000, i.e. up to
125 px); beyond that the kernel walks in 32 bits, and the property test shows why it must."""
import numpy as np
import pytest


def _walk_pk16(e0, A, B, check_range=True):
    a, b = A >> 4, B >> 4                                  # A, B are multiples of 16 (emit_record)
    ep = e0 >> 4                                           # floor(E / 16): E >= 0 <=> floor(E / 16) >= 0
    R = np.array([np.clip(ep, -16000, 16000), np.clip(ep + 4 * b, -16000, 16000)], dtype=np.int64)
    out = 0
    for rp in range(4):
        for x in range(8):
            v = ((R + 32768) % 65536) - 32768             # what a 16-bit half holds
            if check_range:
                assert np.all(v == R), "16-bit overflow inside the walk"
            for h in range(2):
                if v[h] >= 0:
                    out |= 1 << ((rp + 4 * h) * 8 + x)
            if x < 7:
                R = R + a
        R = R + b - 7 * a
    return out


def _walk_exact(e0, A, B):
    out = 0
    for y in range(8):
        for x in range(8):
            if e0 + A * x + B * y >= 0:
                out |= 1 << (y * 8 + x)
    return out


def test_packed_walk_equals_the_integer_rule_for_edges_up_to_125_px():
    rng = np.random.default_rng(16)
    for it in range(60000):
        a = int(rng.integers(-2000, 2001))
        bmax = 2000 - abs(a)
        b = int(rng.integers(-bmax, bmax + 1))
        A, B = 16 * a, 16 * b
        mode = it % 5
        if mode == 0:
            e0 = int(rng.integers(-2**31, 2**31 - 1 - 16 * 16000))          # far from the tile: clamped
        elif mode == 1:
            e0 = int(rng.integers(-300000, 300000))                          # near: partly clamped
        elif mode == 2:                                                      # the edge passes exactly through a pixel (+- the fill-rule bias)
            x, y = int(rng.integers(0, 8)), int(rng.integers(0, 8))
            e0 = -(A * x + B * y) + int(rng.integers(-20, 20))
        elif mode == 3:
            e0 = int(rng.integers(-20, 20))
        else:                                                                # the largest admissible slopes
            a = int(rng.choice([-2000, 2000, 0])); b = int(np.sign(rng.integers(-1, 2)) * (2000 - abs(a)))
            A, B = 16 * a, 16 * b
            e0 = int(rng.integers(-16 * 40000, 16 * 40000))
        assert _walk_pk16(e0, A, B) == _walk_exact(e0, A, B), (e0, A, B)


def test_longer_edges_need_the_32_bit_walk():
    """Beyond the bound the clamped start no longer keeps its distance from zero: some edge function is mis-classified (or
    overflows), which is why a wave that holds such an edge takes the 32-bit walk."""
    rng = np.random.default_rng(17)
    bad = 0
    for _ in range(4000):
        a = int(rng.integers(-20000, 20001)); b = int(rng.integers(-20000, 20001))
        if abs(a) + abs(b) <= 4000:
            continue
        e0 = int(rng.integers(-16 * 200000, 16 * 200000))
        bad += _walk_pk16(e0, 16 * a, 16 * b, check_range=False) != _walk_exact(e0, 16 * a, 16 * b)
    assert bad > 0
