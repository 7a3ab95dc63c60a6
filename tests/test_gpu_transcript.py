"""FiatShamirTranscript's hash as the provers run it on the device (zkhip_transcript_challenge: message schedule on the sixteen lanes
of a row, state rounds on six lanes -- csrc/transcript.hpp) against hashlib and the published SHA-256 vectors, and the transcript's
commit / challenge chain (transcripts/fiat-shamir/src/fiat_shamir.rs:10-40) against the host mirror."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


NIST = [   # FIPS 180-4 / NIST CAVP examples
    (b"abc", "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
    (b"", "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
    (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
    (b"abcdefghbcdefghicdefghijdefghijkefghijklfghijklmghijklmnhijklmnoijklmnopjklmnopqklmnopqrlmnopqrsmnopqrstnopqrstu",
     "cf5b16a778af8380036ce59e7b0492370b249b11e8f07a51afac45037afee9d1"),
]


def test_device_sha256_published_vectors(zk):
    for msg, want in NIST:
        t = zk.DeviceFiatShamirTranscript()
        t.commit(msg)
        assert t.challenge().hex() == want
    t = zk.DeviceFiatShamirTranscript()
    t.commit(b"a" * 1_000_000)                                  # 15 626 blocks through the one wave
    assert t.challenge().hex() == "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"


def test_device_sha256_every_length_across_the_padding_boundaries(zk):
    rng = np.random.default_rng(2024)
    for n in list(range(0, 200)) + [255, 256, 257, 511, 512, 513, 1000, 4095, 4096, 4097]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        t = zk.DeviceFiatShamirTranscript()
        t.commit(data)
        assert t.challenge() == hashlib.sha256(data).digest(), n


def test_device_transcript_chain_matches_the_host_mirror(zk):
    """commit / challenge sequences: a challenge() re-seeds the hasher with its digest (fiat_shamir.rs:21-25); field elements by
    from_be_bytes_mod_order (:27-29)"""
    rng = np.random.default_rng(7)
    dev, host = zk.DeviceFiatShamirTranscript(), zk.FiatShamirTranscript()
    for step in range(60):
        for _ in range(int(rng.integers(0, 4))):
            data = rng.integers(0, 256, int(rng.integers(0, 150)), dtype=np.uint8).tobytes()
            dev.commit(data)
            host.commit(data)
        if step % 3 == 2:
            assert np.array_equal(dev.evaluate_challenge_into_field(), host.evaluate_challenge_into_field())
        else:
            assert dev.challenge() == host.challenge()
    a, b = dev.evaluate_n_challenge_into_field(5), host.evaluate_n_challenge_into_field(5)
    assert np.array_equal(a, b)
