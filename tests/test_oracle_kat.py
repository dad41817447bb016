"""Pins the CPU oracle's bit-level pieces with standards-level known answers and
exhaustive code properties (the reference holds no vectors)."""
import zlib

import numpy as np
import pytest


def test_crc32_matches_ieee(oracle):
    L = oracle.lib()
    msg = np.frombuffer(b"123456789", np.uint8).copy()
    assert L.ll_crc_generate_key(oracle.CRC_32, msg.ctypes.data, 9) == 0xCBF43926
    rng = np.random.RandomState(0)
    for n in (0, 1, 7, 64, 1200):
        m = rng.randint(0, 256, max(n, 1)).astype(np.uint8)
        assert L.ll_crc_generate_key(oracle.CRC_32, m.ctypes.data, n) == zlib.crc32(bytes(m[:n]))


def test_hamming128_corrects_every_single_error(oracle):
    L = oracle.lib()
    words = [L.ll_hamming128_encode_symbol(s) for s in range(256)]
    assert len(set(words)) == 256 and max(words) < 4096
    # minimum distance 3
    dmin = min(bin(words[a] ^ words[b]).count("1") for a in range(256) for b in range(a))
    assert dmin == 3
    for s in range(256):
        assert L.ll_hamming128_decode_symbol(words[s]) == s
        for bit in range(12):
            assert L.ll_hamming128_decode_symbol(words[s] ^ (1 << bit)) == s


def test_golay2412_is_extended_golay(oracle):
    L = oracle.lib()
    words = np.array([L.ll_golay2412_encode_symbol(s) for s in range(4096)], np.uint32)
    wt = np.array([bin(int(w)).count("1") for w in words])
    # weight enumerator of the extended binary Golay code: 1, 759, 2576, 759, 1
    assert {int(k): int(v) for k, v in zip(*np.unique(wt, return_counts=True))} == \
        {0: 1, 8: 759, 12: 2576, 16: 759, 24: 1}
    assert all((int(w) & 0xfff) == s for s, w in enumerate(words))      # systematic
    rng = np.random.RandomState(1)
    for _ in range(3000):
        s = int(rng.randint(0, 4096))
        nerr = int(rng.randint(0, 4))
        e = 0
        for b in rng.choice(24, nerr, replace=False):
            e |= 1 << int(b)
        assert L.ll_golay2412_decode_symbol(int(words[s]) ^ e) == s


def test_small_hamming_codes_are_the_published_tables_and_correct_single_errors(oracle):
    """liquid fec_hamming74.c / fec_hamming84.c generator tables (hamming74_enc_gentab / hamming84_enc_gentab) as
    published; (7,4) is perfect (every 7-bit word within distance 1 of exactly one codeword), (8,4) is its extension."""
    L = oracle.lib()
    h74 = [L.ll_hamming74_encode_symbol(s) for s in range(16)]
    h84 = [L.ll_hamming84_encode_symbol(s) for s in range(16)]
    assert h74 == [0x00, 0x69, 0x2a, 0x43, 0x4c, 0x25, 0x66, 0x0f, 0x70, 0x19, 0x5a, 0x33, 0x3c, 0x55, 0x16, 0x7f]
    assert h84 == [0x00, 0xd2, 0x55, 0x87, 0x99, 0x4b, 0xcc, 0x1e, 0xe1, 0x33, 0xb4, 0x66, 0x78, 0xaa, 0x2d, 0xff]
    assert min(bin(h74[a] ^ h74[b]).count("1") for a in range(16) for b in range(a)) == 3
    assert min(bin(h84[a] ^ h84[b]).count("1") for a in range(16) for b in range(a)) == 4
    for scheme, words, nb in ((oracle.FEC_HAMMING74, h74, 7), (oracle.FEC_HAMMING84, h84, 8)):
        for hi in range(16):
            for lo in range(16):
                msg = np.array([(hi << 4) | lo], np.uint8)
                enc = np.zeros(2, np.uint8)
                L.ll_fec_encode(scheme, 1, msg.ctypes.data, enc.ctypes.data)
                word = (int(enc[0]) << 8) | int(enc[1])
                assert word == ((words[hi] << (16 - nb)) | (words[lo] << (16 - 2 * nb)))     # high nibble first, bit-packed
                for bit in range(16 - 2 * nb, 16):                                       # every single error in either symbol
                    bad = word ^ (1 << bit)
                    e = np.array([bad >> 8, bad & 0xff], np.uint8)
                    dec = np.zeros(1, np.uint8)
                    L.ll_fec_decode(scheme, 1, e.ctypes.data, dec.ctypes.data)
                    assert dec[0] == msg[0]
    # soft decision: an erased bit plus a weak wrong bit in one (8,4) symbol -- two hard errors, which the hard decoder cannot
    # resolve -- decodes by soft distance
    for s in range(16):
        c = h84[s]
        soft = np.array([255 if (c >> (7 - k)) & 1 else 0 for k in range(8)] * 2, np.uint8)
        soft[0] = 127 + (1 if soft[0] == 0 else -1) * 8          # weakly wrong
        soft[3] = 127 + (1 if soft[3] == 0 else 0)               # (nearly) erased, wrong side
        dec = np.zeros(1, np.uint8)
        L.ll_fec_decode_soft(oracle.FEC_HAMMING84, 1, soft.ctypes.data, dec.ctypes.data)
        assert dec[0] == ((s << 4) | s)


@pytest.mark.parametrize("scheme,R", [(2, 3), (3, 5)])
def test_repeat_codes_majority_vote(oracle, scheme, R):
    """liquid fec_rep3.c / fec_rep5.c: R copies in a row, bitwise majority; soft = mean of the copies against 127."""
    L = oracle.lib()
    rng = np.random.RandomState(R)
    n = 37
    msg = rng.randint(0, 256, n).astype(np.uint8)
    enc = np.zeros(R * n, np.uint8)
    assert L.ll_fec_enc_len(scheme, n) == R * n
    L.ll_fec_encode(scheme, n, msg.ctypes.data, enc.ctypes.data)
    assert np.array_equal(enc, np.tile(msg, R))
    # any (R - 1) / 2 copies may be arbitrarily wrong
    bad = enc.copy().reshape(R, n)
    for col in range(n):
        for r in rng.choice(R, (R - 1) // 2, replace=False):
            bad[r, col] = rng.randint(0, 256)
    dec = np.zeros(n, np.uint8)
    L.ll_fec_decode(scheme, n, bad.ctypes.data, dec.ctypes.data)
    assert np.array_equal(dec, msg)
    # soft: the floor of the mean of the copies' soft bits, strictly above 127
    soft = rng.randint(0, 256, (R, n, 8)).astype(np.uint8)
    want = np.packbits((soft.astype(np.int64).sum(0) // R > 127).astype(np.uint8), axis=1)[:, 0]
    L.ll_fec_decode_soft(scheme, n, soft.ctypes.data, dec.ctypes.data)
    assert np.array_equal(dec, want)


@pytest.mark.parametrize("scheme,n", [(6, 1), (6, 2), (6, 7), (6, 1204), (7, 1), (7, 2), (7, 3), (7, 18), (7, 1204), (1, 5),
                                      (2, 1), (2, 1204), (3, 9), (3, 1204), (4, 1), (4, 3), (4, 4), (4, 1204), (5, 1), (5, 1204)])
def test_fec_block_roundtrip_and_lengths(oracle, scheme, n):
    L = oracle.lib()
    rng = np.random.RandomState(n)
    msg = rng.randint(0, 256, n).astype(np.uint8)
    k = L.ll_fec_enc_len(scheme, n)
    bits_in = 8 * n
    m_bits, k_bits = {6: (8, 12), 7: (12, 24), 1: (8, 8), 2: (8, 24), 3: (8, 40), 4: (4, 7), 5: (4, 8)}[scheme]
    blocks = -(-bits_in // m_bits)
    assert k == -(-(blocks * k_bits) // 8)                 # liquid fec_block_get_enc_msg_len
    enc = np.zeros(k, np.uint8)
    L.ll_fec_encode(scheme, n, msg.ctypes.data, enc.ctypes.data)
    dec = np.zeros(n, np.uint8)
    L.ll_fec_decode(scheme, n, enc.ctypes.data, dec.ctypes.data)
    assert np.array_equal(dec, msg)
    soft = np.repeat(enc, 8).reshape(-1, 8)
    soft = ((soft >> (7 - np.arange(8))) & 1).astype(np.uint8) * 255
    dec2 = np.zeros(n, np.uint8)
    L.ll_fec_decode_soft(scheme, n, soft.ctypes.data, dec2.ctypes.data)
    assert np.array_equal(dec2, msg)


def test_hamming128_soft_beats_hard_on_weak_double_error(oracle):
    L = oracle.lib()
    s = 0xA7
    c = L.ll_hamming128_encode_symbol(s)
    bits = np.array([(c >> (11 - k)) & 1 for k in range(12)])
    soft = np.where(bits == 1, 230, 25).astype(np.uint8)
    soft[3] = 140 if bits[3] == 0 else 115        # two weak (barely wrong) bits
    soft[9] = 140 if bits[9] == 0 else 115
    enc_soft = np.concatenate([soft, np.full(4, 0, np.uint8)])      # one symbol = 2 bytes = 16 soft bits
    out = np.zeros(1, np.uint8)
    L.ll_fec_decode_soft(6, 1, enc_soft.ctypes.data, out.ctypes.data)
    assert out[0] == s


@pytest.mark.parametrize("n", [2, 3, 36, 37, 100, 1806, 2409])
def test_interleaver_is_a_bit_permutation(oracle, n):
    L = oracle.lib()
    rng = np.random.RandomState(n)
    x = rng.randint(0, 256, n).astype(np.uint8)
    for depth in range(5):
        y = np.zeros(n, np.uint8)
        z = np.zeros(n, np.uint8)
        L.ll_interleaver_encode(n, depth, x.ctypes.data, y.ctypes.data)
        L.ll_interleaver_decode(n, depth, y.ctypes.data, z.ctypes.data)
        assert np.array_equal(z, x)
        assert int(np.unpackbits(y).sum()) == int(np.unpackbits(x).sum())
        if depth == 0:
            assert np.array_equal(y, x)
        # soft de-interleaver moves soft bits exactly like the hard one moves bits
        soft = np.unpackbits(y).astype(np.uint8) * 200 + 20
        out = np.zeros(8 * n, np.uint8)
        L.ll_interleaver_decode_soft(n, depth, soft.ctypes.data, out.ctypes.data)
        assert np.array_equal(np.packbits(out > 127), x)
    if n >= 36:
        y = np.zeros(n, np.uint8)
        L.ll_interleaver_encode(n, 4, x.ctypes.data, y.ctypes.data)
        assert not np.array_equal(y, x)


def test_scrambler_is_involution(oracle):
    L = oracle.lib()
    x = np.arange(37, dtype=np.uint8)
    y = x.copy()
    L.ll_scramble(y.ctypes.data, 37)
    assert np.array_equal(y[:4] ^ x[:4], np.array([0xb4, 0x6a, 0x8b, 0xc5], np.uint8))
    L.ll_scramble(y.ctypes.data, 37)
    assert np.array_equal(y, x)


def test_conv_v27_is_the_k7_rate_half_code(oracle):
    """liquid LIQUID_FEC_CONV_V27 (libfec viterbi27): generators 0x6d / 0x4f (octal 155 / 117 read the other way round:
    the standard K = 7 pair with free distance 10), 2 (8 n + 6) output bits, and a maximum-likelihood decoder: any 4 bit
    errors in a block are corrected (d_free = 10), soft decisions decode what the hard slicer cannot."""
    L = oracle.lib()
    n = 40
    rng = np.random.RandomState(7)
    msg = rng.randint(0, 256, n).astype(np.uint8)
    k = L.ll_fec_enc_len(11, n)
    assert k == 2 * n + 2                                  # ceil(2 (8 n + 6) / 8): liquid fec_conv_get_enc_msg_len
    enc = np.zeros(k, np.uint8)
    L.ll_fec_encode(11, n, msg.ctypes.data, enc.ctypes.data)
    # impulse response of the encoder = the two generator polynomials, MSB = newest bit
    one = np.zeros(2, np.uint8); one[0] = 0x80
    e1 = np.zeros(L.ll_fec_enc_len(11, 2), np.uint8)
    L.ll_fec_encode(11, 2, one.ctypes.data, e1.ctypes.data)
    bits = np.unpackbits(e1)[:14].reshape(7, 2)
    assert int("".join(map(str, bits[:, 0])), 2) == 0b1011011 and int("".join(map(str, bits[:, 1])), 2) == 0b1111001   # 0x6d, 0x4f reversed
    assert bits.sum() == 10                                # d_free of the (171, 133) code
    dec = np.zeros(n, np.uint8)
    for trial in range(20):
        bad = np.unpackbits(enc)
        pos = rng.choice(len(bad) - 16, 4, replace=False)
        bad[pos] ^= 1
        b = np.packbits(bad)
        L.ll_fec_decode(11, n, b.ctypes.data, dec.ctypes.data)
        assert np.array_equal(dec, msg), trial
    # soft: three adjacent symbols erased to 127/128 plus two confident errors nearby -- decodes; sliced hard it is 5 errors in a span
    soft = (np.unpackbits(enc).astype(np.int32) * 255)
    soft[100:103] = 127 + (soft[100:103] > 0)             # (nearly) erased, sliced to the right side
    soft[110] = 255 - soft[110]; soft[117] = 255 - soft[117]
    sb = soft.astype(np.uint8)
    L.ll_fec_decode_soft(11, n, sb.ctypes.data, dec.ctypes.data)
    assert np.array_equal(dec, msg)


@pytest.mark.parametrize("n,fec0,fec1,enc", [(14, 7, 1, 36), (1200, 1, 6, 1806), (1200, 1, 7, 2409),
                                             (1200, 1, 1, 1204), (0, 1, 6, 6), (5, 6, 7, 30),
                                             (1200, 1, 11, 2410), (100, 11, 6, 315), (0, 1, 11, 10),
                                             (1200, 1, 2, 3612), (1200, 1, 3, 6020), (1200, 1, 4, 2107), (1200, 1, 5, 2408),
                                             (33, 4, 2, 195), (33, 5, 3, 370), (21, 2, 4, 132), (7, 3, 5, 110)])
def test_packetizer_lengths_roundtrip_and_error_correction(oracle, n, fec0, fec1, enc):
    p = oracle.Packetizer(n, oracle.CRC_32, fec0, fec1)
    assert p.enc_len == enc
    rng = np.random.RandomState(n + fec1)
    msg = rng.randint(0, 256, n).astype(np.uint8)
    pkt = p.encode(bytes(msg))
    ok, out = p.decode(pkt)
    assert ok and np.array_equal(out, msg)
    soft = (np.unpackbits(pkt) * 255).astype(np.uint8)
    ok, out = p.decode_soft(soft)
    assert ok and np.array_equal(out, msg)
    if fec1 != 1 and enc >= 30:
        bad = pkt.copy()
        for pos in rng.choice(enc, 3, replace=False):
            bad[pos] ^= 1 << int(rng.randint(0, 8))       # interleaving spreads these over codewords
        ok, out = p.decode(bad)
        assert ok and np.array_equal(out, msg)
    if n:
        bad = pkt.copy()
        bad[:] ^= 0xff
        ok, _ = p.decode(bad)
        assert not ok


def test_modems_gray_and_soft_bits(oracle):
    for scheme, M in ((oracle.MODEM_BPSK, 2), (oracle.MODEM_QPSK, 4), (oracle.MODEM_QAM16, 16), (oracle.MODEM_QAM64, 64)):
        m = oracle.Modem(scheme)
        pts = np.array([m.modulate(s) for s in range(M)])
        assert abs(np.mean(np.abs(pts) ** 2) - 1.0) < 1e-6                 # unit average energy
        d = np.abs(pts[:, None] - pts[None, :])
        dmin = d[d > 1e-6].min()
        for s in range(M):
            assert m.demodulate(pts[s]) == s
            s2, soft = m.demodulate_soft(pts[s])
            assert s2 == s
            bits = [(s >> (m.bps - 1 - k)) & 1 for k in range(m.bps)]
            assert all((sb > 127) == bool(b) for sb, b in zip(soft, bits))
            for t in range(M):                                              # Gray: nearest neighbours differ in 1 bit
                if t != s and d[s, t] < dmin * 1.01:
                    assert bin(s ^ t).count("1") == 1
    q = oracle.Modem(oracle.MODEM_QPSK)
    s, soft = q.demodulate_soft(np.complex64(0.01 - 0.7j))
    assert s == 2 and soft[0] == 255 and 120 <= soft[1] <= 127
