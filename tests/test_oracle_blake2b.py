"""Pins the oracle's own BLAKE2b against RFC 7693 appendix A and Python's hashlib (libsodium-equivalent)."""
import hashlib

import numpy as np

import oracle


def test_rfc7693_abc():
    exp = ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
           "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    assert oracle.blake2b(b"abc", outlen=64).hex() == exp


def test_matches_hashlib_unkeyed_and_keyed():
    rng = np.random.default_rng(7)
    for ln in [0, 1, 24, 32, 48, 63, 64, 96, 127, 128, 129, 192, 255, 256, 257, 576, 1000]:
        data = bytes(rng.integers(0, 256, size=ln, dtype=np.uint8))
        for outlen in (8, 24, 32, 64):
            assert oracle.blake2b(data, outlen) == hashlib.blake2b(data, digest_size=outlen).digest()
            key = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))
            assert oracle.blake2b(data, outlen, key) == hashlib.blake2b(data, digest_size=outlen, key=key).digest()


def test_hashchain_quirk_and_squeeze():
    # absorb() hashes only the first 32 bytes of state||input -> state' = H(state)
    hc = oracle.Hashchain()
    assert bytes(hc.state) == b" " * 32
    hc.absorb(b"\x01" * 32)
    s1 = bytes(hc.state)
    assert s1 == hashlib.blake2b(b" " * 32, digest_size=32).digest()
    assert s1.hex().startswith("e468d42e")      # value recorded by the survey's probe of the reference
    hc2 = oracle.Hashchain()
    hc2.absorb(b"\xff" * 32)
    assert bytes(hc2.state) == s1
    # squeeze: keyed BLAKE2b(state || index_le64, key = i_le64, 24 bytes), raw into the words
    x = hc.squeeze(2, 3)
    for i in range(2):
        exp = hashlib.blake2b(s1 + (1).to_bytes(8, "little"), digest_size=24, key=i.to_bytes(8, "little")).digest()
        assert x[i].tobytes() == exp
    pos = hc.squeeze_query_positions(3, 1 << 10)
    for k, p in enumerate(pos):
        d = hashlib.blake2b(s1, digest_size=8, key=(2 + k).to_bytes(8, "little")).digest()
        assert p == int.from_bytes(d, "little") % (1 << 10)
