"""liquid_usrp_amd -- MI355X-native drop-in for liquid-usrp's multichannel OFDM receive path.

This package is the Python face of ``lib/libmcrx_hip.so`` (hand-written gfx950 kernels behind
the C-ABI of ``include/mcrx_hip.h``).  :class:`multichannelrx` mirrors the reference C++ class
(``include/multichannelrx.h:29-83``): same constructor arguments, ``Execute`` / ``Reset`` /
``GetNumChannels``, per-channel callbacks with the ``framesync_callback`` argument list.

There is no CPU implementation here: if the HIP library or a GPU is missing, construction
raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never imported by
this package.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("MCRX_LIB") or os.path.join(_HERE, "lib", "libmcrx_hip.so")      # (MCRX_LIB: A/B builds)

MCRX_OK, MCRX_EINVAL, MCRX_ENOMEM, MCRX_EHIP, MCRX_EUNSUPP, MCRX_EOVERFLOW, MCRX_EBUSY = 0, -1, -2, -3, -4, -5, -6
TILE = 16         # MCRX_TILE (include/mcrx_hip.h)
TX_TILE = 8       # granules of the transmit side (mctx_hip_traffic_tiles)

LIQUID_CRC_NONE, LIQUID_CRC_32 = 1, 6
LIQUID_FEC_NONE, LIQUID_FEC_HAMMING128, LIQUID_FEC_GOLAY2412, LIQUID_FEC_CONV_V27 = 1, 6, 7, 11
LIQUID_FEC_REP3, LIQUID_FEC_REP5, LIQUID_FEC_HAMMING74, LIQUID_FEC_HAMMING84 = 2, 3, 4, 5
LIQUID_MODEM_QAM16, LIQUID_MODEM_QAM64, LIQUID_MODEM_BPSK, LIQUID_MODEM_QPSK = 27, 29, 39, 40


def build(force=False):
    """Compile the HIP library for gfx950 with the in-tree Makefile (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".h", ".hpp"))]
    deps.append(os.path.join(_HERE, "..", "include", "mcrx_hip.h"))
    stale = (not os.path.exists(_LIBPATH)) or any(os.path.getmtime(d) > os.path.getmtime(_LIBPATH) for d in deps)
    if force or stale:
        subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return _LIBPATH


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_payload_len", C.c_uint32), ("max_frames", C.c_uint32),
                ("payload_soft", C.c_uint32), ("slab_blocks", C.c_uint32), ("channel_first", C.c_uint32),
                ("channel_count", C.c_uint32), ("batch_samples", C.c_uint32), ("single_channel", C.c_uint32),
                ("serial", C.c_uint32), ("chunk_blocks", C.c_uint32), ("defer_samples", C.c_uint32), ("front_end", C.c_uint32), ("skip_framesyms", C.c_uint32),
                ("worker_build", C.c_uint32), ("acquisition", C.c_uint32), ("scout_build", C.c_uint32), ("conv_scratch", C.c_uint32)]


class FrameC(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("header_valid", C.c_int32), ("payload_valid", C.c_int32),
                ("payload_len", C.c_uint32), ("header", C.c_uint8 * 8),
                ("evm", C.c_float), ("rssi", C.c_float), ("cfo", C.c_float),
                ("mod_scheme", C.c_uint32), ("mod_bps", C.c_uint32), ("check", C.c_uint32),
                ("fec0", C.c_uint32), ("fec1", C.c_uint32), ("num_framesyms", C.c_uint32),
                ("end_sample", C.c_uint64), ("payload", C.c_void_p), ("framesyms", C.c_void_p)]


_EXPORTS = {
    "mcrx_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]),
    "mcrx_hip_destroy": (C.c_int, [C.c_void_p]),
    "mcrx_hip_reset": (C.c_int, [C.c_void_p]),
    "mcrx_hip_reset_at": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mcrx_hip_num_channels": (C.c_uint, [C.c_void_p]),
    "mcrx_hip_execute_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mcrx_hip_execute_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mcrx_hip_flush": (C.c_int, [C.c_void_p]),
    "mcrx_hip_poll": (C.c_int, [C.c_void_p]),
    "mcrx_hip_discard": (C.c_int, [C.c_void_p]),
    "mcrx_hip_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcrx_hip_launches": (C.c_uint64, [C.c_void_p]),
    "mcrx_hip_history_tiles": (C.c_uint, [C.c_void_p]),
    "mcrx_hip_stream_wait_launch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "mcrx_hip_spec_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "mcrx_hip_viterbi_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "mcrx_hip_frames_pending": (C.c_size_t, [C.c_void_p]),
    "mcrx_hip_next_frame": (C.c_int, [C.c_void_p, C.POINTER(FrameC)]),
    "mcrx_hip_drain_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mcrx_hip_frames_dropped": (C.c_uint64, [C.c_void_p]),
    "mcrx_hip_channelize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]),
    "mcrx_hip_sync": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p]),
    "mcrx_hip_restart": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcrx_hip_get_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mcrx_hip_nco_step": (C.c_uint32, [C.c_void_p]),
    "mcrx_hip_history_blocks": (C.c_uint, [C.c_void_p]),
    "mcrx_hip_device": (C.c_int, [C.c_void_p]),
    "mcrx_hip_selftest_device_table": (C.c_int, []),
    "mcrx_hip_kernel_time_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mcrx_hip_kernel_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "mcrx_hip_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mcrx_hip_last_error": (C.c_char_p, []),
    "msresamp_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_float, C.c_float]),
    "msresamp_hip_destroy": (C.c_int, [C.c_void_p]),
    "msresamp_hip_reset": (C.c_int, [C.c_void_p]),
    "msresamp_hip_get_delay": (C.c_float, [C.c_void_p]),
    "msresamp_hip_max_output": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "msresamp_hip_execute_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_size_t), C.c_void_p]),
    "msresamp_hip_last_error": (C.c_char_p, []),
    "mcrx_hip_pfb2_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_float]),
    "mcrx_hip_pfb2_destroy": (C.c_int, [C.c_void_p]),
    "mcrx_hip_pfb2_get_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mcrx_hip_pfb2_analyze": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mcrx_hip_pfb2_last_error": (C.c_char_p, []),
    "mcrx_hip_pipeline_unique_id": (C.c_int, [C.c_void_p]),
    "mcrx_hip_pipeline_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_uint]),
    "mcrx_hip_pipeline_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcrx_hip_pipeline_push_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcrx_hip_pipeline_wait": (C.c_int, [C.c_void_p]),
    "mcrx_hip_pipeline_host_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mcrx_hip_pipeline_reset": (C.c_int, [C.c_void_p, C.c_int64]),
    "mcrx_hip_pipeline_comm_count": (C.c_int, [C.c_void_p]),
    "mcrx_hip_pipeline_time_exchange": (C.c_int, [C.c_void_p, C.c_int]),
    "mcrx_hip_pipeline_exchange_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "mcrx_hip_pipeline_bytes_sent_per_round": (C.c_uint64, [C.c_void_p]),
    "mcrx_hip_pipeline_destroy": (C.c_int, [C.c_void_p]),
    "mcrx_hip_pipeline_last_error": (C.c_char_p, []),
    "mctx_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p]),
    "mctx_hip_destroy": (C.c_int, [C.c_void_p]),
    "mctx_hip_blocks_for": (C.c_size_t, [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]),
    "mctx_hip_generate_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                                           C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "mctx_hip_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mctx_hip_traffic_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int,
                                          C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mctx_hip_traffic_destroy": (C.c_int, [C.c_void_p]),
    "mctx_hip_traffic_tiles": (C.c_int, [C.c_void_p, C.c_longlong, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mctx_hip_synthesize_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_longlong, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_float, C.c_void_p, C.c_void_p]),
    "mctx_hip_frame_len": (C.c_size_t, [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]),
    "mctx_hip_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_void_p, C.c_size_t]),
    "mctx_hip_stream_begin": (C.c_int, [C.c_void_p, C.c_uint]),
    "mctx_hip_stream_ready": (C.c_int, [C.c_void_p, C.c_uint]),
    "mctx_hip_stream_update": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]),
    "mctx_hip_stream_generate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mctx_hip_stream_reset": (C.c_int, [C.c_void_p]),
    "mctx_hip_last_error": (C.c_char_p, []),
}

_lib = None


def exported_symbols():
    return sorted(_EXPORTS)


def lib():
    """Load libmcrx_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise RuntimeError("libmcrx_hip.so is not built (run __graft_entry__.build()); "
                               "liquid_usrp_amd has no CPU fallback")
        try:
            # PyTorch-ROCm bundles its own HIP runtime; load it first so this process holds a
            # single libamdhip64 (device buffers and streams are shared with torch tensors).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIBPATH)
        for name, (res, args) in _EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class McrxError(RuntimeError):
    pass


def _check(rc, allow=()):
    if rc != MCRX_OK and rc not in allow:
        msg = lib().mcrx_hip_last_error()
        raise McrxError("mcrx_hip error %d: %s" % (rc, msg.decode() if msg else ""))
    return rc


class Frame(object):
    """Arguments of one framesync_callback invocation (include/multichannelrx.h:45)."""
    __slots__ = ("channel", "header", "header_valid", "payload", "payload_valid", "evm", "rssi", "cfo",
                 "framesyms", "mod_scheme", "mod_bps", "check", "fec0", "fec1", "end_sample")

    def __repr__(self):
        return "Frame(ch=%d hv=%d pv=%d len=%d evm=%.2f end=%d)" % (
            self.channel, self.header_valid, self.payload_valid, len(self.payload), self.evm, self.end_sample)


def _dptr(t):
    """Raw device pointer of a torch tensor / int / None."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


class multichannelrx(object):
    """GPU multichannel OFDM receiver with the reference class's interface.

    multichannelrx(num_channels, M, cp_len, taper_len, p, userdata, callback)
      p         subcarrier allocation (bytes / uint8 array) or None for the default
      userdata  list of per-channel objects handed back to the callbacks (or None)
      callback  list of per-channel callables
                cb(header, header_valid, payload, payload_len, payload_valid, stats, userdata) -> int
                (stats is the :class:`Frame`: evm, rssi, cfo, framesyms, ...)
    Callbacks run on the calling thread at flush points (buffer full, Flush(), Reset(), close()),
    in the reference's order (frame end time, then channel).
    """

    def __init__(self, num_channels, M, cp_len, taper_len, p=None, userdata=None, callback=None, **cfg):
        self._h = C.c_void_p()
        self.N, self.K, self.M, self.cp = num_channels, 2 * num_channels, M, cp_len
        parr = None if p is None else np.ascontiguousarray(np.frombuffer(bytes(bytearray(p)), np.uint8))
        c = Config()
        c.struct_size = C.sizeof(Config)
        c.payload_soft = 1
        for k, v in cfg.items():
            setattr(c, k, v)
        rc = lib().mcrx_hip_create(C.byref(self._h), num_channels, M, cp_len, taper_len,
                                   None if parr is None else parr.ctypes.data, C.addressof(c))
        if rc != MCRX_OK:
            self._h = C.c_void_p()
            msg = lib().mcrx_hip_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)           # the reference prints the message and throws
            raise McrxError("mcrx_hip_create failed (%d): %s" % (rc, msg))
        self.userdata = list(userdata) if userdata is not None else [None] * num_channels
        self.callback = list(callback) if callback is not None else [None] * num_channels
        self.frames = []            # every frame delivered so far (also handed to the callbacks)

    # ---- reference API -----------------------------------------------------------------
    def GetNumChannels(self):
        return self.N

    def Execute(self, x, num_samples=None, stream=None):
        """Push samples: a complex64 numpy array (host) or a torch complex64 CUDA tensor (HBM)."""
        if hasattr(x, "is_cuda") and x.is_cuda:
            n = int(x.numel()) if num_samples is None else int(num_samples)
            _check(lib().mcrx_hip_execute_device(self._h, _dptr(x), n, _stream_ptr(stream)))
            return
        a = np.ascontiguousarray(x, np.complex64)
        n = a.size if num_samples is None else int(num_samples)
        _check(lib().mcrx_hip_execute_host(self._h, a.ctypes.data, n), allow=(MCRX_EOVERFLOW,))   # drops are counted
        self._deliver(flush=False)

    def Reset(self):
        _check(lib().mcrx_hip_reset(self._h))
        self._deliver(flush=False)

    # ---- additions -----------------------------------------------------------------------
    def Flush(self):
        """Process everything pushed so far and deliver the callbacks."""
        rc = lib().mcrx_hip_flush(self._h)
        _check(rc, allow=(MCRX_EOVERFLOW,))
        self._deliver(flush=False)
        return rc

    def Poll(self, deliver=True):
        """Overlapped harvest (mcrx_hip_poll): frames of everything pushed before the previous Poll."""
        rc = lib().mcrx_hip_poll(self._h)
        _check(rc, allow=(MCRX_EOVERFLOW,))
        if deliver:
            self._deliver(flush=False)
        return rc

    def Discard(self):
        _check(lib().mcrx_hip_discard(self._h))

    def frames_pending(self):
        return int(lib().mcrx_hip_frames_pending(self._h))

    def drain_count(self):
        """Walk the harvested frames through the C-ABI (mcrx_hip_drain_count: the loop a C caller writes around
        mcrx_hip_next_frame) without building Python objects; returns (frames, valid payloads, payload bytes)."""
        n, ok, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().mcrx_hip_drain_count(self._h, C.byref(n), C.byref(ok), C.byref(nb)))
        return int(n.value), int(ok.value), int(nb.value)

    def stream_wait(self, stream=None, launch=None):
        if launch is None:
            _check(lib().mcrx_hip_stream_wait(self._h, _stream_ptr(stream)))
        else:
            _check(lib().mcrx_hip_stream_wait_launch(self._h, launch, _stream_ptr(stream)))

    def spec_stats(self, reset=False):
        """(frames acquired by the scouts' own walk, frames adopted from speculative waves)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().mcrx_hip_spec_stats(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    def viterbi_stats(self, reset=False):
        """(frames through the K = 7 decoder's own kernel, forward passes repeated, traceback passes repeated)"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().mcrx_hip_viterbi_stats(self._h, C.byref(a), C.byref(b), C.byref(c), 1 if reset else 0))
        return int(a.value), int(b.value), int(c.value)

    def _deliver(self, flush):
        f = FrameC()
        while lib().mcrx_hip_next_frame(self._h, C.byref(f)) == 1:
            fr = Frame()
            fr.channel = int(f.channel)
            fr.header = bytes(bytearray(f.header))
            fr.header_valid, fr.payload_valid = int(f.header_valid), int(f.payload_valid)
            fr.payload = C.string_at(f.payload, f.payload_len) if f.payload and f.payload_len else b""
            fr.evm, fr.rssi, fr.cfo = float(f.evm), float(f.rssi), float(f.cfo)
            if f.framesyms and f.num_framesyms:
                buf = (C.c_float * (2 * f.num_framesyms)).from_address(f.framesyms)
                fr.framesyms = np.frombuffer(buf, np.float32).copy().view(np.complex64)
            else:
                fr.framesyms = np.zeros(0, np.complex64)
            fr.mod_scheme, fr.mod_bps = int(f.mod_scheme), int(f.mod_bps)
            fr.check, fr.fec0, fr.fec1 = int(f.check), int(f.fec0), int(f.fec1)
            fr.end_sample = int(f.end_sample)
            self.frames.append(fr)
            cb = self.callback[fr.channel] if fr.channel < len(self.callback) else None
            if cb is not None:
                cb(fr.header, fr.header_valid, fr.payload, len(fr.payload), fr.payload_valid, fr,
                   self.userdata[fr.channel])

    # ---- stage level (bench / multi-GPU / parity tests) ------------------------------------
    def taps(self):
        h = np.zeros(14 * self.K, np.float32)
        _check(lib().mcrx_hip_get_taps(self._h, h.ctypes.data, h.size))
        return h

    def nco_step(self):
        return int(lib().mcrx_hip_nco_step(self._h))

    def history_blocks(self):
        """Blocks of 2N samples of filter history the analysis bank needs in front of a push (13; 27 with front_end = 1)."""
        return int(lib().mcrx_hip_history_blocks(self._h))

    def channelize(self, d_iq, nblocks, first_sample, d_out, groups=1, d_halo=None, stream=None):
        _check(lib().mcrx_hip_channelize(self._h, _dptr(d_iq), nblocks, first_sample, _dptr(d_halo),
                                         _dptr(d_out), groups, _stream_ptr(stream)))

    def sync(self, d_chan, first_sample, nsamples, stream=None):
        """Returns the launch number (for stream_wait).  first_sample may be negative (history in front of sample 0)."""
        _check(lib().mcrx_hip_sync(self._h, _dptr(d_chan), first_sample & 0xFFFFFFFFFFFFFFFF, nsamples, _stream_ptr(stream)))
        return int(lib().mcrx_hip_launches(self._h)) - 1

    @property
    def hist_tiles(self):
        """tiles of channel-rate history a stage-level sync needs in front of new samples"""
        return int(lib().mcrx_hip_history_tiles(self._h))

    def restart(self, stream=None):
        _check(lib().mcrx_hip_restart(self._h, _stream_ptr(stream)))

    def kernel_time_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().mcrx_hip_kernel_time_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_timing(self, on=True):
        """per-kernel HIP-event timing on / off (off in a new receiver); returns the previous setting"""
        return bool(lib().mcrx_hip_kernel_timing(self._h, 1 if on else 0))

    def kernel_stats(self, reset=False):
        """{kernel: (total_ms, launches)} from HIP events recorded on the launch stream while kernel_timing() was on."""
        names = ("channelizer_kernel", "sync_kernel", "place_jobs_kernel", "payload_kernel", "decode_kernel")
        ms, cnt = (C.c_double * len(names))(), (C.c_uint64 * len(names))()
        _check(lib().mcrx_hip_kernel_stats(self._h, ms, cnt, 1 if reset else 0))
        return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}

    def frames_dropped(self):
        return int(lib().mcrx_hip_frames_dropped(self._h))

    def close(self):
        if self._h:
            try:
                self.Flush()
            finally:
                lib().mcrx_hip_destroy(self._h)
                self._h = C.c_void_p()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mcrx_hip_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class msresamp(object):
    """GPU mirror of liquid's msresamp_crcf as the reference front ends use it
    (src/flexframe_rx.cc:179,240): msresamp(rate, As); execute(x) -> y, with x / y torch
    complex64 CUDA tensors (IQ stays in HBM).  rate <= 1 decimates (receive front ends), rate > 1 interpolates
    (the transmit applications' msresamp_crcf_create(2.0, 60), src/flexframe_tx.cc:170)."""

    def __init__(self, rate, As=60.0):
        self._h = C.c_void_p()
        rc = lib().msresamp_hip_create(C.byref(self._h), rate, As)
        if rc != MCRX_OK:
            self._h = C.c_void_p()
            msg = lib().msresamp_hip_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)
            raise McrxError("msresamp_hip_create failed (%d): %s" % (rc, msg))
        self.rate = rate

    def get_delay(self):
        return float(lib().msresamp_hip_get_delay(self._h))

    def reset(self):
        lib().msresamp_hip_reset(self._h)

    def execute(self, x, stream=None):
        import torch
        n = int(x.numel())
        cap = int(lib().msresamp_hip_max_output(self._h, n)) + 8
        y = torch.empty(cap, dtype=torch.complex64, device=x.device)
        nout = C.c_size_t(0)
        rc = lib().msresamp_hip_execute_device(self._h, _dptr(x), n, _dptr(y), cap, C.byref(nout), _stream_ptr(stream))
        if rc != MCRX_OK:
            raise McrxError("msresamp_hip_execute_device failed (%d): %s" % (rc, lib().msresamp_hip_last_error().decode()))
        return y[:nout.value]

    def close(self):
        if self._h:
            lib().msresamp_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TxTraffic(object):
    """The frames of one channel shard of a multichanneltx, modulated and resident in HBM (mctx_hip_traffic_*).
    sent[c] = [(header, payload), ...] for local channel c (global channel = channel_first + c)."""

    def __init__(self, tx, channel_first, channel_count, frames, payload_len, mod, fec0, fec1, seed, stream=None):
        import torch
        self._t = C.c_void_p()
        self.tx, self.channel_first, self.channel_count = tx, channel_first, channel_count
        hdr = np.zeros((channel_count, frames, 8), np.uint8)
        pay = np.zeros((channel_count, frames, max(payload_len, 1)), np.uint8)
        st = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().mctx_hip_traffic_create(tx._h, C.byref(self._t), channel_first, channel_count, frames, payload_len, mod, fec0,
                                           fec1, seed & 0xFFFFFFFF, hdr.ctypes.data, pay.ctypes.data, _stream_ptr(st))
        if rc != MCRX_OK:
            self._t = C.c_void_p()
            msg = lib().mctx_hip_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)
            raise McrxError("mctx_hip_traffic_create failed (%d): %s" % (rc, msg))
        self.blocks = int(lib().mctx_hip_blocks_for(tx._h, frames, payload_len, mod, fec0, fec1))
        self.sent = [[(bytes(hdr[c, f]), bytes(pay[c, f, :payload_len])) for f in range(frames)] for c in range(channel_count)]

    def tiles(self, first_block, nblocks, out, stream=None):
        """Channel-rate granules out[tile][c][8] of blocks [first_block, first_block+nblocks) (zeros outside the traffic)."""
        import torch
        assert nblocks % 8 == 0 and out.numel() >= nblocks * self.channel_count
        st = stream if stream is not None else torch.cuda.current_stream(out.device)
        rc = lib().mctx_hip_traffic_tiles(self._t, first_block, nblocks, _dptr(out), _stream_ptr(st))
        if rc != MCRX_OK:
            raise McrxError("mctx_hip_traffic_tiles failed (%d): %s" % (rc, lib().mctx_hip_last_error().decode()))
        return out

    def close(self):
        if getattr(self, "_t", None) is not None and self._t:
            lib().mctx_hip_traffic_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class pipeline(object):
    """The C-ABI multi-GPU receive pipeline (mcrx_hip_pipeline_*: csrc/pipeline.hip) -- sharding.Pipeline's schedule without
    Python between the stages, the exchange as grouped ncclSend / ncclRecv.  `rx` is this rank's receiver handle (its channel
    shard, defer_samples set); world > 1: `unique_id` = the 128 bytes rank 0 got from pipeline.unique_id()."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        rc = lib().mcrx_hip_pipeline_unique_id(buf)
        if rc != MCRX_OK:
            raise McrxError("mcrx_hip_pipeline_unique_id failed (%d): %s" % (rc, lib().mcrx_hip_pipeline_last_error().decode()))
        return bytes(buf)

    def __init__(self, rx, rank, world, sub_blocks, unique_id=None, nbuf=3):
        self._h = C.c_void_p()
        self.rx, self.rank, self.world, self.Tc = rx, rank, world, sub_blocks
        uid = None if unique_id is None else (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(lib().mcrx_hip_pipeline_create(C.byref(self._h), rx._h, rank, world, uid, sub_blocks, nbuf), "create")
        self.rounds = 0

    def _chk(self, rc, what):
        if rc != MCRX_OK:
            raise McrxError("mcrx_hip_pipeline_%s failed (%d): %s" % (what, rc, lib().mcrx_hip_pipeline_last_error().decode()))

    def push(self, iq_sub, halo=None, after=None, ready=False):
        """`after`: the torch stream that produced iq_sub / halo (default: the current one) -- the round starts behind what is enqueued
        there.  ready=True: the buffers are complete (the caller has synchronized since writing them), wait for nothing
        (MCRX_STREAM_READY).  Mind that torch's default stream is HIP's legacy NULL stream: an event recorded there is a barrier
        across every blocking stream of the process, the handle's own included -- a caller that pushes from the default stream
        every round serializes its rounds (bench.py --pipeline: 170 -> 70 Gsample/s); push from a side stream, or say ready."""
        import torch
        if ready:
            ptr = C.c_void_p(-1 & 0xFFFFFFFFFFFFFFFF)
        else:
            st = after if after is not None else torch.cuda.current_stream(iq_sub.device)
            ptr = _stream_ptr(st)
        self._chk(lib().mcrx_hip_pipeline_push(self._h, _dptr(iq_sub), _dptr(halo), ptr), "push")
        self.rounds += 1

    def wait(self):
        self._chk(lib().mcrx_hip_pipeline_wait(self._h), "wait")

    def time_exchange(self, on=True):
        self._chk(lib().mcrx_hip_pipeline_time_exchange(self._h, 1 if on else 0), "time_exchange")

    def exchange_ms(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(lib().mcrx_hip_pipeline_exchange_ms(self._h, C.byref(ms), C.byref(n), 1 if reset else 0), "exchange_ms")
        return ms.value, int(n.value)

    def bytes_sent_per_round(self):
        return int(lib().mcrx_hip_pipeline_bytes_sent_per_round(self._h))

    def comm_count(self):
        """Ranks of the RCCL communicator behind the exchange, as ncclCommCount reports them (1 at world 1; -1: call missing)."""
        return int(lib().mcrx_hip_pipeline_comm_count(self._h))

    def reset(self, extra_samples=0):
        """multichannelrx::Reset() for the sharded receiver (every rank, same point of the stream): mcrx_hip_pipeline_reset."""
        self._chk(lib().mcrx_hip_pipeline_reset(self._h, int(extra_samples)), "reset")
        self.rounds = 0

    def close(self):
        if self._h:
            lib().mcrx_hip_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class multichanneltx(object):
    """GPU multichannel OFDM transmitter used as the synthetic IQ source
    (reference: lib/multichanneltx.cc + the traffic loop of src/multichannel_tx.cc:163-213).

    generate(frames_per_channel, payload_len, ...) -> (iq, sent): iq is a torch complex64 CUDA
    tensor with the wideband stream, sent[ch] = [(header, payload), ...]."""

    def __init__(self, num_channels, M, cp_len, taper_len, p=None, max_payload_len=2048):
        self._h = C.c_void_p()
        self.N, self.K, self.M, self.cp = num_channels, 2 * num_channels, M, cp_len
        self.max_payload_len, self._stream_max = max_payload_len, -1
        parr = None if p is None else np.ascontiguousarray(np.frombuffer(bytes(bytearray(p)), np.uint8))
        rc = lib().mctx_hip_create(C.byref(self._h), num_channels, M, cp_len, taper_len,
                                   None if parr is None else parr.ctypes.data)
        if rc != MCRX_OK:
            self._h = C.c_void_p()
            msg = lib().mctx_hip_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)
            raise McrxError("mctx_hip_create failed (%d): %s" % (rc, msg))

    def GetNumChannels(self):
        return self.N

    def generate(self, frames_per_channel, payload_len, mod=LIQUID_MODEM_QPSK, fec0=LIQUID_FEC_NONE,
                 fec1=LIQUID_FEC_HAMMING128, gain=None, seed=0xC0FFEE, nblocks=None, device=None):
        import torch
        nb = int(lib().mctx_hip_blocks_for(self._h, frames_per_channel, payload_len, mod, fec0, fec1))
        if nblocks is not None:
            nb = max(nb, (int(nblocks) + TILE - 1) // TILE * TILE)
        iq = torch.empty(nb * self.K, dtype=torch.complex64, device=device or "cuda")
        hdr = np.zeros((self.N, frames_per_channel, 8), np.uint8)
        pay = np.zeros((self.N, frames_per_channel, max(payload_len, 1)), np.uint8)
        g = (1.0 / self.N) if gain is None else gain
        stream = torch.cuda.current_stream(iq.device)
        rc = lib().mctx_hip_generate(self._h, _dptr(iq), nb, frames_per_channel, payload_len, mod, fec0, fec1, g,
                                     seed & 0xFFFFFFFF, hdr.ctypes.data, pay.ctypes.data, _stream_ptr(stream))
        if rc != MCRX_OK:
            raise McrxError("mctx_hip_generate failed (%d): %s" % (rc, lib().mctx_hip_last_error().decode()))
        sent = [[(bytes(hdr[c, f]), bytes(pay[c, f, :payload_len])) for f in range(frames_per_channel)]
                for c in range(self.N)]
        return iq, sent

    def generate_ragged(self, nblocks, len_lo=64, len_hi=1200, gap_max=3, long_every=8, long_max=184, mod=LIQUID_MODEM_QPSK,
                        fec0=LIQUID_FEC_NONE, fec1=LIQUID_FEC_HAMMING128, gain=None, seed=0xC0FFEE, device=None):
        """Ragged traffic (src/multichannel_txrx.cc:227-267): frames of len_lo .. len_hi bytes after 0 .. gap_max idle symbols,
        a longer silence (16 .. 16 + long_max symbols) once in long_every frames, until `nblocks` blocks are full.
        -> (iq, sent, starts): sent[ch] = [(header, payload)], starts[ch] = [block index of each frame's first sample]."""
        import torch
        nb = (int(nblocks) + TILE - 1) // TILE * TILE
        L = self.M + self.cp
        shortest = int(lib().mctx_hip_blocks_for(self._h, 1, len_lo, mod, fec0, fec1)) - 64
        maxf = nb // max(shortest, L) + 2
        iq = torch.empty(nb * self.K, dtype=torch.complex64, device=device or "cuda")
        cnt = np.zeros(self.N, np.uint32); hdr = np.zeros((self.N, maxf, 8), np.uint8); ln = np.zeros((self.N, maxf), np.uint32)
        pay = np.zeros((self.N, maxf, max(len_hi, 1)), np.uint8); start = np.zeros((self.N, maxf), np.uint64)
        g = (1.0 / self.N) if gain is None else gain
        stream = torch.cuda.current_stream(iq.device)
        rc = lib().mctx_hip_generate_ragged(self._h, _dptr(iq), nb, maxf, len_lo, len_hi, gap_max, long_every, long_max, mod, fec0, fec1,
                                            g, seed & 0xFFFFFFFF, cnt.ctypes.data, hdr.ctypes.data, ln.ctypes.data, pay.ctypes.data,
                                            start.ctypes.data, _stream_ptr(stream))
        if rc != MCRX_OK:
            raise McrxError("mctx_hip_generate_ragged failed (%d): %s" % (rc, lib().mctx_hip_last_error().decode()))
        sent = [[(bytes(hdr[c, f]), bytes(pay[c, f, :ln[c, f]])) for f in range(int(cnt[c]))] for c in range(self.N)]
        starts = [[int(start[c, f]) for f in range(int(cnt[c]))] for c in range(self.N)]
        return iq, sent, starts

    # ---- sharded form: channel-sharded frame generators, time-sharded synthesis bank (see sharding.TxPipeline)
    def traffic(self, channel_first, channel_count, frames_per_channel, payload_len, mod=LIQUID_MODEM_QPSK,
                fec0=LIQUID_FEC_NONE, fec1=LIQUID_FEC_HAMMING128, seed=0xC0FFEE, stream=None):
        """Frames of a channel shard (same recipe and seeds as generate()): a TxTraffic with .tiles() and .sent."""
        return TxTraffic(self, channel_first, channel_count, frames_per_channel, payload_len, mod, fec0, fec1, seed, stream)

    def synthesize(self, tiles, groups, first_block, nblocks, lead_blocks, keep_blocks=0, gain=None, out=None, stream=None):
        """Wideband samples of blocks [first_block-keep, first_block+nblocks) from exchanged channel-rate granules
        tiles[groups][(lead+nblocks)/8][N/groups][8] (mctx_hip_synthesize_tiles)."""
        import torch
        if out is None:
            out = torch.empty((keep_blocks + nblocks) * self.K, dtype=torch.complex64, device=tiles.device)
        assert tiles.numel() >= (lead_blocks + nblocks) * self.N and out.numel() >= (keep_blocks + nblocks) * self.K
        g = (1.0 / self.N) if gain is None else gain
        st = stream if stream is not None else torch.cuda.current_stream(tiles.device)
        self._chk(lib().mctx_hip_synthesize_tiles(self._h, _dptr(tiles), groups, first_block, nblocks, lead_blocks, keep_blocks,
                                                  g, _dptr(out), _stream_ptr(st)), "mctx_hip_synthesize_tiles")
        return out

    # ---- class interface of the reference (lib/multichanneltx.cc:126-227), served by the GPU one symbol period at a time
    def _begin(self, payload_len):
        if self._stream_max < 0:
            self._stream_max = max(int(self.max_payload_len), int(payload_len))
            self._chk(lib().mctx_hip_stream_begin(self._h, self._stream_max), "mctx_hip_stream_begin")

    def _chk(self, rc, what):
        if rc != MCRX_OK:
            msg = lib().mctx_hip_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)
            raise McrxError("%s failed (%d): %s" % (what, rc, msg))

    def Reset(self):
        if self._stream_max >= 0:
            self._chk(lib().mctx_hip_stream_reset(self._h), "mctx_hip_stream_reset")

    def IsChannelReadyForData(self, channel_id):
        if not 0 <= channel_id < self.N:
            raise ValueError("error: multichanneltx::IsChannelReadyForData(), invalid channel id")
        if self._stream_max < 0:
            return True
        return lib().mctx_hip_stream_ready(self._h, channel_id) == 1

    def UpdateData(self, channel_id, header, payload, mod=LIQUID_MODEM_QPSK, fec0=LIQUID_FEC_NONE, fec1=LIQUID_FEC_HAMMING128):
        """Returns False (the reference prints a warning and returns) when the channel is busy."""
        if not 0 <= channel_id < self.N:
            raise ValueError("error: multichanneltx::UpdateData(), invalid channel id")
        h = np.frombuffer(bytes(bytearray(header))[:8].ljust(8, b"\0"), np.uint8)
        pl = np.frombuffer(bytes(bytearray(payload)), np.uint8)
        self._begin(len(pl))
        rc = lib().mctx_hip_stream_update(self._h, channel_id, h.ctypes.data, pl.ctypes.data if len(pl) else None, len(pl), mod, fec0, fec1)
        if rc == MCRX_EBUSY:
            return False
        self._chk(rc, "mctx_hip_stream_update")
        return True

    def GenerateSamples(self, buffer=None):
        """The next 2N wideband samples (host numpy complex64; written into `buffer` when given)."""
        self._begin(0)
        out = np.empty(self.K, np.complex64) if buffer is None else buffer
        assert out.dtype == np.complex64 and out.size >= self.K and out.flags.c_contiguous
        self._chk(lib().mctx_hip_stream_generate(self._h, out.ctypes.data), "mctx_hip_stream_generate")
        return out

    def frame(self, header, payload, mod=LIQUID_MODEM_QPSK, fec0=LIQUID_FEC_NONE, fec1=LIQUID_FEC_HAMMING128, gain=1.0):
        """All samples of one frame of one frame generator (ofdmflexframegen assemble + writesymbol to the last
        symbol, lib/ofdmtxrx.cc:297-342), channel rate, host numpy complex64."""
        h = np.frombuffer(bytes(bytearray(header))[:8].ljust(8, b"\0"), np.uint8)
        pl = np.frombuffer(bytes(bytearray(payload)), np.uint8)
        n = int(lib().mctx_hip_frame_len(self._h, len(pl), mod, fec0, fec1))
        if n == 0:
            raise ValueError("unsupported frame properties")
        out = np.empty(n, np.complex64)
        self._chk(lib().mctx_hip_frame(self._h, h.ctypes.data, pl.ctypes.data if len(pl) else None, len(pl), mod, fec0, fec1,
                                       gain, out.ctypes.data, n), "mctx_hip_frame")
        return out

    def close(self):
        if self._h:
            lib().mctx_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class firpfbch2(object):
    """2x-oversampled analysis bank on the GPU (liquid firpfbch2_crcf analyzer): M channels, M/2 samples per step.

    analyze(x) takes a torch complex64 CUDA tensor holding a whole number of steps and returns [nsteps, M];
    the filter state carries over between calls (the last 2*m*M samples are kept in HBM)."""

    def __init__(self, num_channels, m, As=60.0):
        self._h = C.c_void_p()
        self.M, self.m = num_channels, m
        rc = lib().mcrx_hip_pfb2_create(C.byref(self._h), num_channels, m, As)
        if rc != MCRX_OK:
            self._h = C.c_void_p()
            msg = lib().mcrx_hip_pfb2_last_error().decode()
            if rc == MCRX_EINVAL:
                raise ValueError(msg)
            raise McrxError("mcrx_hip_pfb2_create failed (%d): %s" % (rc, msg))
        self._hist = None
        self._step = 0

    def taps(self):
        h = np.zeros(2 * self.m * self.M, np.float32)
        _check(lib().mcrx_hip_pfb2_get_taps(self._h, h.ctypes.data, h.size))
        return h

    def reset(self):
        self._hist, self._step = None, 0

    def analyze(self, x):
        import torch
        M2, depth = self.M // 2, 2 * self.m * self.M
        n = int(x.numel())
        if n % M2:
            raise ValueError("input must be a whole number of M/2-sample steps")
        ns = n // M2
        lead = 0 if self._hist is None else int(self._hist.numel())
        buf = x if lead == 0 else torch.cat([self._hist, x])
        out = torch.empty((ns, self.M), dtype=torch.complex64, device=x.device)
        stream = torch.cuda.current_stream(x.device)
        rc = lib().mcrx_hip_pfb2_analyze(self._h, C.c_void_p(buf.data_ptr() + 8 * lead), lead, ns, self._step, _dptr(out),
                                         _stream_ptr(stream))
        if rc != MCRX_OK:
            raise McrxError("mcrx_hip_pfb2_analyze failed (%d): %s" % (rc, lib().mcrx_hip_pfb2_last_error().decode()))
        self._hist = buf[-depth:].clone() if buf.numel() > depth else buf.clone()
        self._step += ns
        return out

    def close(self):
        if self._h:
            lib().mcrx_hip_pfb2_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ofdmflexframesync(multichannelrx):
    """One frame synchronizer fed its channel's samples directly -- the receive half of the reference's
    ofdmtxrx (lib/ofdmtxrx.cc:91,620-626) -- on the same kernels as the multichannel bank, without a channelizer.
    callback(header, header_valid, payload, payload_len, payload_valid, stats, userdata)."""

    def __init__(self, M, cp_len, taper_len, p=None, callback=None, userdata=None, **cfg):
        multichannelrx.__init__(self, 1, M, cp_len, taper_len, p, [userdata], [callback], single_channel=1, **cfg)
        self.K = 1

    execute = multichannelrx.Execute
    reset = multichannelrx.Reset


def ofdmflexframegen(M, cp_len, taper_len, p=None):
    """One frame generator (the transmit half of ofdmtxrx): .frame(header, payload, mod, fec0, fec1, gain)."""
    return multichanneltx(1, M, cp_len, taper_len, p)


def tiles_to_channels(chan, nch):
    """[tile][ch][TILE] (torch or numpy, complex) -> [ch][time] numpy array (test helper)."""
    a = chan.cpu().numpy() if hasattr(chan, "cpu") else np.asarray(chan)
    a = a.reshape(-1, nch, TILE)
    return np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(nch, -1)
