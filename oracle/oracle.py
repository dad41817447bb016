"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/liquidlite.h ("PARITY UNPINNED").  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
if os.environ.get("LL_ORACLE_LIB"):      # the D6 / D7 variants (oracle/Makefile: variants; tests/test_gpu_variants.py)
    _LIB = os.environ["LL_ORACLE_LIB"]

CRC_NONE, CRC_32 = 1, 6
FEC_NONE, FEC_HAMMING128, FEC_GOLAY2412, FEC_CONV_V27 = 1, 6, 7, 11
FEC_REP3, FEC_REP5, FEC_HAMMING74, FEC_HAMMING84 = 2, 3, 4, 5
MODEM_QAM16, MODEM_QAM64, MODEM_BPSK, MODEM_QPSK = 27, 29, 39, 40
ANALYZER, SYNTHESIZER = 0, 1


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc)."""
    srcs = [f for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB) for f in srcs)
    if os.environ.get("LL_ORACLE_LIB"):
        return _LIB                                      # (a variant: built by `make -C oracle variants`)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB


class Stats(C.Structure):
    _fields_ = [("evm", C.c_float), ("rssi", C.c_float), ("cfo", C.c_float),
                ("framesyms", C.c_void_p), ("num_framesyms", C.c_uint),
                ("mod_scheme", C.c_uint), ("mod_bps", C.c_uint), ("check", C.c_uint),
                ("fec0", C.c_uint), ("fec1", C.c_uint)]


class CF(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Props(C.Structure):
    _fields_ = [("check", C.c_uint), ("fec0", C.c_uint), ("fec1", C.c_uint), ("mod_scheme", C.c_uint)]


FRAMESYNC_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_ubyte), C.c_uint,
                           C.c_int, Stats, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    vp, u, f, i = C.c_void_p, C.c_uint, C.c_float, C.c_int

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("ll_besseli0", f, f)
    sig("ll_firdes_kaiser", None, u, f, f, f, vp)
    sig("ll_fft", None, u, vp, vp, i)
    sig("ll_nco_rad2u32", C.c_uint32, f)
    sig("ll_nco_sincos_u32", None, C.c_uint32, vp, vp)
    sig("ll_msequence_init_default", None, vp, u)
    sig("ll_msequence_advance", u, vp)
    sig("ll_firpfbch_create_kaiser", vp, i, u, u, f)
    sig("ll_firpfbch_destroy", None, vp)
    sig("ll_firpfbch_reset", None, vp)
    sig("ll_firpfbch_analyzer_execute", None, vp, vp, vp)
    sig("ll_firpfbch_synthesizer_execute", None, vp, vp, vp)
    sig("ll_firpfbch_get_taps", u, vp, vp)
    sig("ll_modem_create", vp, i)
    sig("ll_modem_destroy", None, vp)
    sig("ll_modem_bps", u, vp)
    sig("ll_modem_modulate", CF, vp, u)
    sig("ll_modem_demodulate", u, vp, CF)
    sig("ll_modem_demodulate_soft", u, vp, CF, vp)
    sig("ll_modem_get_evm", f, vp)
    sig("ll_modem_soft_neighbors", vp, vp, vp)
    sig("ll_crc_generate_key", u, i, vp, u)
    sig("ll_fec_enc_len", u, i, u)
    sig("ll_fec_encode", None, i, u, vp, vp)
    sig("ll_fec_decode", None, i, u, vp, vp)
    sig("ll_fec_decode_soft", None, i, u, vp, vp)
    sig("ll_hamming128_encode_symbol", u, u)
    sig("ll_hamming128_decode_symbol", u, u)
    sig("ll_hamming74_encode_symbol", u, u)
    sig("ll_hamming84_encode_symbol", u, u)
    sig("ll_fec_supported", i, i)
    sig("ll_golay2412_encode_symbol", u, u)
    sig("ll_golay2412_decode_symbol", u, u)
    sig("ll_interleaver_encode", None, u, u, vp, vp)
    sig("ll_interleaver_decode", None, u, u, vp, vp)
    sig("ll_interleaver_decode_soft", None, u, u, vp, vp)
    sig("ll_scramble", None, vp, u)
    sig("ll_packetizer_create", vp, u, i, i, i)
    sig("ll_packetizer_destroy", None, vp)
    sig("ll_packetizer_enc_len", u, vp)
    sig("ll_packetizer_encode", None, vp, vp, vp)
    sig("ll_packetizer_decode", i, vp, vp, vp)
    sig("ll_packetizer_decode_soft", i, vp, vp, vp)
    sig("ll_ofdmframe_init_default_sctype", None, u, vp)
    sig("ll_ofdmframe_validate_sctype", i, vp, u, vp, vp, vp)
    sig("ll_ofdmframe_init_S0", None, vp, u, vp, vp, vp)
    sig("ll_ofdmframe_init_S1", None, vp, u, vp, vp, vp)
    sig("ll_ofdmframe_eq_smoother", None, vp, u, u, vp)
    sig("ll_ofdmframe_pilot_fit", None, vp, u, vp)
    sig("ll_ofdmflexframegen_create", vp, u, u, u, vp, vp)
    sig("ll_ofdmflexframegen_destroy", None, vp)
    sig("ll_ofdmflexframegen_setprops", None, vp, vp)
    sig("ll_ofdmflexframegen_getframelen", u, vp)
    sig("ll_ofdmflexframegen_assemble", None, vp, vp, vp, u)
    sig("ll_ofdmflexframegen_writesymbol", i, vp, vp)
    sig("ll_ofdmflexframesync_create", vp, u, u, u, vp, FRAMESYNC_CB, vp)
    sig("ll_ofdmflexframesync_destroy", None, vp)
    sig("ll_ofdmflexframesync_reset", None, vp)
    sig("ll_ofdmflexframesync_set_soft", None, vp, i)
    sig("ll_ofdmflexframesync_execute", None, vp, vp, u)
    sig("ll_msresamp_create", vp, f, f)
    sig("ll_msresamp_destroy", None, vp)
    sig("ll_msresamp_execute", None, vp, vp, u, vp, vp)
    sig("ll_mcrx_create", vp, u, u, u, u, vp, vp, vp)
    sig("ll_mcrx_destroy", None, vp)
    sig("ll_mcrx_reset", None, vp)
    sig("ll_mcrx_set_soft", None, vp, i)
    sig("ll_mcrx_execute", None, vp, vp, u)
    sig("ll_mcrx_set_front_end", None, vp, C.c_int)
    sig("ll_mcrx_channelize_oversampled", None, vp, vp, u, vp)
    sig("ll_mcrx_execute_parallel", None, vp, vp, u, i)
    sig("ll_mcrx_channelize", None, vp, vp, u, vp)
    sig("ll_mctx_create", vp, u, u, u, u, vp)
    sig("ll_mctx_destroy", None, vp)
    sig("ll_mctx_is_channel_ready", i, vp, u)
    sig("ll_mctx_update_data", i, vp, u, vp, vp, u, i, i, i)
    sig("ll_mctx_generate_samples", None, vp, vp)
    sig("ll_mctx_reset", None, vp)
    sig("ll_firpfbch2_create_kaiser", vp, u, u, f)
    sig("ll_firpfbch2_destroy", None, vp)
    sig("ll_firpfbch2_reset", None, vp)
    sig("ll_firpfbch2_get_taps", u, vp, vp)
    sig("ll_firpfbch2_analyze", None, vp, vp, u, vp)
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _bytes(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


# ---------------------------------------------------------------- small helpers
def firdes_kaiser(n, fc, As, mu=0.0):
    h = np.zeros(n, np.float32)
    lib().ll_firdes_kaiser(n, fc, As, mu, _ptr(h))
    return h


def fft(x, backward=False):
    x = np.ascontiguousarray(x, np.complex64)
    y = np.zeros_like(x)
    lib().ll_fft(len(x), _ptr(x), _ptr(y), 1 if backward else 0)
    return y


def default_sctype(M):
    p = np.zeros(M, np.uint8)
    lib().ll_ofdmframe_init_default_sctype(M, _ptr(p))
    return p


def init_S0S1(p):
    M = len(p)
    out = {}
    for name in ("S0", "S1"):
        S = np.zeros(M, np.complex64)
        s = np.zeros(M, np.complex64)
        cnt = C.c_uint(0)
        getattr(lib(), "ll_ofdmframe_init_" + name)(_ptr(p), M, _ptr(S), _ptr(s), C.byref(cnt))
        out[name] = (S, s, cnt.value)
    return out


def eq_smoother(p, order=4):
    M = len(p)
    nen = int(np.count_nonzero(p))
    S = np.zeros((M, nen), np.float32)
    lib().ll_ofdmframe_eq_smoother(_ptr(p), M, order, _ptr(S))
    return S


def pilot_fit(p):
    M = len(p)
    npil = int(np.count_nonzero(p == 1))
    P = np.zeros((2, npil), np.float32)
    lib().ll_ofdmframe_pilot_fit(_ptr(p), M, _ptr(P))
    return P


class Modem:
    def __init__(self, scheme):
        self.q = lib().ll_modem_create(scheme)
        self.bps = lib().ll_modem_bps(self.q)

    def modulate(self, sym):
        r = lib().ll_modem_modulate(self.q, int(sym))
        return np.complex64(complex(r.re, r.im))

    def demodulate(self, x):
        return lib().ll_modem_demodulate(self.q, CF(float(np.real(x)), float(np.imag(x))))

    def demodulate_soft(self, x):
        soft = np.zeros(8, np.uint8)
        s = lib().ll_modem_demodulate_soft(self.q, CF(float(np.real(x)), float(np.imag(x))), _ptr(soft))
        return s, soft[:self.bps].copy()

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_modem_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class MsResamp:
    """Oracle msresamp_crcf (decimating for rate < 1, interpolating for rate > 1)."""

    def __init__(self, rate, As=60.0):
        self.rate = rate
        self.q = lib().ll_msresamp_create(rate, As)
        if not self.q:
            raise ValueError("rate must be positive")

    def execute(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        y = np.zeros(int(len(x) * self.rate * 1.001) + 64, np.complex64)
        ny = C.c_uint(0)
        lib().ll_msresamp_execute(self.q, _ptr(x), len(x), _ptr(y), C.byref(ny))
        return y[:ny.value].copy()

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_msresamp_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class Channelizer2:
    """2x-oversampled analysis bank (liquid firpfbch2_crcf analyzer): M channels, M/2 samples per step."""

    def __init__(self, M, m, As=60.0):
        self.M = M
        self.q = lib().ll_firpfbch2_create_kaiser(M, m, As)
        if not self.q:
            raise ValueError("invalid firpfbch2 arguments")

    def taps(self):
        n = lib().ll_firpfbch2_get_taps(self.q, None)
        h = np.zeros(n, np.float32)
        lib().ll_firpfbch2_get_taps(self.q, _ptr(h))
        return h

    def analyze(self, x):
        """x: whole number of M/2-sample steps -> [nsteps, M]; state carries over between calls."""
        x = np.ascontiguousarray(x, np.complex64)
        ns = len(x) // (self.M // 2)
        y = np.zeros((ns, self.M), np.complex64)
        lib().ll_firpfbch2_analyze(self.q, _ptr(x), ns, _ptr(y))
        return y

    def reset(self):
        lib().ll_firpfbch2_reset(self.q)

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_firpfbch2_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class Channelizer:
    def __init__(self, kind, K, m, As=60.0):
        self.K = K
        self.q = lib().ll_firpfbch_create_kaiser(kind, K, m, As)

    def taps(self):
        n = lib().ll_firpfbch_get_taps(self.q, None)
        h = np.zeros(n, np.float32)
        lib().ll_firpfbch_get_taps(self.q, _ptr(h))
        return h

    def analyze(self, x):
        x = np.ascontiguousarray(x, np.complex64).reshape(-1, self.K)
        y = np.zeros_like(x)
        for b in range(x.shape[0]):
            lib().ll_firpfbch_analyzer_execute(self.q, _ptr(x[b]), _ptr(y[b]))
        return y

    def synthesize(self, X):
        X = np.ascontiguousarray(X, np.complex64).reshape(-1, self.K)
        y = np.zeros_like(X)
        for b in range(X.shape[0]):
            lib().ll_firpfbch_synthesizer_execute(self.q, _ptr(X[b]), _ptr(y[b]))
        return y

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_firpfbch_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class Packetizer:
    def __init__(self, n, crc=CRC_32, fec0=FEC_NONE, fec1=FEC_NONE):
        self.n = n
        self.q = lib().ll_packetizer_create(n, crc, fec0, fec1)
        self.enc_len = lib().ll_packetizer_enc_len(self.q)

    def encode(self, msg):
        m = _bytes(msg)
        out = np.zeros(self.enc_len, np.uint8)
        lib().ll_packetizer_encode(self.q, _ptr(m), _ptr(out))
        return out

    def decode(self, pkt):
        p = np.ascontiguousarray(pkt, np.uint8)
        out = np.zeros(max(self.n, 1), np.uint8)
        ok = lib().ll_packetizer_decode(self.q, _ptr(p), _ptr(out))
        return bool(ok), out[:self.n]

    def decode_soft(self, soft):
        p = np.ascontiguousarray(soft, np.uint8)
        assert p.size == 8 * self.enc_len
        out = np.zeros(max(self.n, 1), np.uint8)
        ok = lib().ll_packetizer_decode_soft(self.q, _ptr(p), _ptr(out))
        return bool(ok), out[:self.n]

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_packetizer_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class Frame:
    """One decoded frame as delivered to the framesync callback."""
    __slots__ = ("channel", "header", "header_valid", "payload", "payload_valid", "evm", "rssi", "cfo",
                 "framesyms", "mod_scheme", "mod_bps", "check", "fec0", "fec1")

    def __repr__(self):
        return "Frame(ch=%s hv=%d pv=%d len=%d evm=%.2f rssi=%.2f)" % (
            self.channel, self.header_valid, self.payload_valid, len(self.payload), self.evm, self.rssi)


def _make_cb(sink, channel):
    def cb(header, header_valid, payload, payload_len, payload_valid, stats, ud):
        fr = Frame()
        fr.channel = channel
        fr.header = bytes(bytearray(header[i] for i in range(8))) if header else b""
        fr.header_valid = int(header_valid)
        fr.payload = bytes(bytearray(payload[i] for i in range(payload_len))) if payload and payload_len else b""
        fr.payload_valid = int(payload_valid)
        fr.evm, fr.rssi, fr.cfo = stats.evm, stats.rssi, stats.cfo
        if stats.framesyms and stats.num_framesyms:
            buf = (C.c_float * (2 * stats.num_framesyms)).from_address(stats.framesyms)
            fr.framesyms = np.frombuffer(buf, np.float32).copy().view(np.complex64)
        else:
            fr.framesyms = np.zeros(0, np.complex64)
        fr.mod_scheme, fr.mod_bps = stats.mod_scheme, stats.mod_bps
        fr.check, fr.fec0, fr.fec1 = stats.check, stats.fec0, stats.fec1
        sink.append(fr)
        return 0
    return FRAMESYNC_CB(cb)


class FlexFrameGen:
    def __init__(self, M, cp, taper, p=None, check=CRC_32, fec0=FEC_NONE, fec1=FEC_HAMMING128, mod=MODEM_QPSK):
        self.M, self.cp = M, cp
        self._p = None if p is None else np.ascontiguousarray(p, np.uint8)
        props = Props(check, fec0, fec1, mod)
        self.q = lib().ll_ofdmflexframegen_create(M, cp, taper, None if p is None else _ptr(self._p),
                                                  C.addressof(props))

    def setprops(self, check=CRC_32, fec0=FEC_NONE, fec1=FEC_HAMMING128, mod=MODEM_QPSK):
        props = Props(check, fec0, fec1, mod)
        lib().ll_ofdmflexframegen_setprops(self.q, C.addressof(props))

    def frame(self, header, payload):
        """Assemble and return all samples of one frame (including the tail symbol)."""
        h = _bytes(header)
        pl = _bytes(payload) if len(payload) else np.zeros(1, np.uint8)
        lib().ll_ofdmflexframegen_assemble(self.q, _ptr(h), _ptr(pl), len(payload))
        nsym = lib().ll_ofdmflexframegen_getframelen(self.q)
        L = self.M + self.cp
        out = np.zeros(nsym * L, np.complex64)
        for s in range(nsym):
            last = lib().ll_ofdmflexframegen_writesymbol(self.q, _ptr(out[s * L:(s + 1) * L]))
            assert bool(last) == (s == nsym - 1)
        return out

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_ofdmflexframegen_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class FlexFrameSync:
    def __init__(self, M, cp, taper, p=None, soft=True):
        self.frames = []
        self._cb = _make_cb(self.frames, 0)
        self._p = None if p is None else np.ascontiguousarray(p, np.uint8)
        self.q = lib().ll_ofdmflexframesync_create(M, cp, taper, None if p is None else _ptr(self._p),
                                                   self._cb, None)
        lib().ll_ofdmflexframesync_set_soft(self.q, 1 if soft else 0)

    def execute(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        lib().ll_ofdmflexframesync_execute(self.q, _ptr(x), len(x))

    def reset(self):
        lib().ll_ofdmflexframesync_reset(self.q)

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_ofdmflexframesync_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class MultiChannelRx:
    """Oracle mirror of the reference class (lib/multichannelrx.cc)."""

    def __init__(self, N, M, cp, taper, p=None, soft=True, count_only=False, front_end=0):
        """count_only: frames are only counted, in C (self.counts()), instead of being handed to Python --
        for timing the receiver itself.  front_end=1: the 2x-oversampled bank + half-band adapter in place of the
        critically sampled bank."""
        self.N, self.K = N, 2 * N
        self.frames = []
        if count_only:
            self._counter = (C.c_ulonglong * 4)()
            fn = C.cast(lib().ll_counting_callback, C.c_void_p).value
            arr = (C.c_void_p * N)(*([fn] * N))
            ud = (C.c_void_p * N)(*([C.addressof(self._counter)] * N))
        else:
            self._cbs = [_make_cb(self.frames, c) for c in range(N)]
            arr = (FRAMESYNC_CB * N)(*self._cbs)
            ud = (C.c_void_p * N)()
        self._p = None if p is None else np.ascontiguousarray(p, np.uint8)
        self.q = lib().ll_mcrx_create(N, M, cp, taper, None if p is None else _ptr(self._p),
                                      C.cast(ud, C.c_void_p), C.cast(arr, C.c_void_p))
        if not self.q:
            raise ValueError("invalid multichannelrx arguments")
        lib().ll_mcrx_set_soft(self.q, 1 if soft else 0)
        if front_end:
            lib().ll_mcrx_set_front_end(self.q, 1)

    def channelize_oversampled(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        nb = len(x) // self.K
        out = np.zeros((nb, self.N), np.complex64)
        lib().ll_mcrx_channelize_oversampled(self.q, _ptr(x), nb, _ptr(out))
        return out

    def execute(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        lib().ll_mcrx_execute(self.q, _ptr(x), len(x))

    def counts(self):
        """(frames, valid headers, valid payloads, payload bytes) of a count_only receiver."""
        return tuple(int(v) for v in self._counter)

    def execute_parallel(self, x, nthreads):
        """Same frames as execute() (grouped by channel instead of by time), on `nthreads` host threads."""
        x = np.ascontiguousarray(x, np.complex64)
        lib().ll_mcrx_execute_parallel(self.q, _ptr(x), len(x), int(nthreads))

    def channelize(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        nb = len(x) // self.K
        out = np.zeros((nb, self.N), np.complex64)
        lib().ll_mcrx_channelize(self.q, _ptr(x), nb, _ptr(out))
        return out

    def reset(self):
        lib().ll_mcrx_reset(self.q)

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_mcrx_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


class MultiChannelTx:
    """Oracle mirror of the reference class (lib/multichanneltx.cc)."""

    def __init__(self, N, M, cp, taper, p=None):
        self.N, self.K = N, 2 * N
        self._p = None if p is None else np.ascontiguousarray(p, np.uint8)
        self.q = lib().ll_mctx_create(N, M, cp, taper, None if p is None else _ptr(self._p))
        if not self.q:
            raise ValueError("invalid multichanneltx arguments")

    def ready(self, ch):
        return lib().ll_mctx_is_channel_ready(self.q, ch) == 1

    def update(self, ch, header, payload, mod=MODEM_QPSK, fec0=FEC_NONE, fec1=FEC_HAMMING128):
        h, pl = _bytes(header), _bytes(payload)
        return lib().ll_mctx_update_data(self.q, ch, _ptr(h), _ptr(pl), len(payload), mod, fec0, fec1)

    def reset(self):
        lib().ll_mctx_reset(self.q)

    def generate(self, nblocks):
        out = np.zeros((nblocks, self.K), np.complex64)
        for b in range(nblocks):
            lib().ll_mctx_generate_samples(self.q, _ptr(out[b]))
        return out.reshape(-1)

    def __del__(self):
        try:
            if getattr(self, "q", None):
                lib().ll_mctx_destroy(self.q)
                self.q = None
        except Exception:          # interpreter shutdown
            pass


def synth_traffic(N, M, cp, taper, nframes, payload_len=1200, seed=0xC0FFEE, mod=MODEM_QPSK,
                  fec0=FEC_NONE, fec1=FEC_HAMMING128, gain=None, extra_blocks=64, p=None):
    """Wideband IQ following the reference's traffic recipe (src/multichannel_tx.cc:163-213):
    header = [pid_hi, pid_lo, channel, 5 random bytes], random payload, frames back to back on
    every channel, soft gain 1/N.  Returns (iq, sent) with sent[ch] = [(header, payload), ...]."""
    tx = MultiChannelTx(N, M, cp, taper, p)
    rngs = [np.random.RandomState((seed + c) & 0x7FFFFFFF) for c in range(N)]
    sent = [[] for _ in range(N)]
    pid = [0] * N
    g = (1.0 / N) if gain is None else gain
    chunks = []
    L = M + cp
    done = False
    idle_blocks = 0
    while not done:
        for c in range(N):
            if pid[c] < nframes and tx.ready(c):
                hdr = bytes([(pid[c] >> 8) & 0xff, pid[c] & 0xff, c & 0xff]) + bytes(rngs[c].randint(0, 256, 5).astype(np.uint8))
                pl = bytes(rngs[c].randint(0, 256, payload_len).astype(np.uint8))
                tx.update(c, hdr, pl, mod, fec0, fec1)
                sent[c].append((hdr, pl))
                pid[c] += 1
        chunks.append(tx.generate(L))
        if all(pid[c] >= nframes and tx.ready(c) for c in range(N)):
            idle_blocks += L
            done = idle_blocks >= extra_blocks
    iq = np.concatenate(chunks) * np.float32(g)
    return iq.astype(np.complex64), sent
