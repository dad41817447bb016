"""Run by tests/test_gpu_variants.py in a process of its own, with MCRX_LIB and LL_ORACLE_LIB pointing at one PAIR of variant builds
(the kernels and the oracle with the same S1-stage switch flipped: DESIGN.md section 2, D6 / D7).  The GPU receiver must make the
variant oracle's decisions frame for frame -- valid or not -- and match its equalised symbols to 1e-5; how many of the frames that were
sent survive the flipped switch is printed beside it (it is the documented reason the default setting is what it is)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    from __graft_entry__ import load_product, load_oracle
    prod, ora = load_product(), load_oracle()
    from test_gpu_parity import check_frames
    out = {"mcrx_lib": os.path.basename(os.environ.get("MCRX_LIB", "default")), "oracle_lib": os.path.basename(os.environ.get("LL_ORACLE_LIB", "default")), "cases": []}
    cases = [(8, 64, 8, 40, 6, 300, 3, None), (2, 64, 8, 40, 6, 1200, 2, None), (4, 64, 8, 27, 7, 200, 3, 25.0), (2, 256, 32, 27, 7, 400, 2, None)]
    for N, M, cp, mod, fec1, plen, nf, snr in cases:
        iq, sent = ora.synth_traffic(N, M, cp, 4, nf, payload_len=plen, mod=mod, fec1=fec1, seed=5)
        x = iq[:len(iq) // (32 * N) * (32 * N)].astype(np.complex64)
        if snr is not None:
            rng = np.random.RandomState(2)
            nstd = np.sqrt(np.mean(np.abs(x) ** 2)) * 10.0 ** (-snr / 20.0) / np.sqrt(2.0)
            x = (x + nstd * (rng.randn(len(x)) + 1j * rng.randn(len(x)))).astype(np.complex64)
        o = ora.MultiChannelRx(N, M, cp, 4)
        o.execute(x)
        rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen)
        half = len(x) // 2 // (32 * N) * (32 * N)
        rx.Execute(x[:half]); rx.Execute(x[half:]); rx.Flush()
        case = {"N": N, "M": M, "mod": mod, "fec1": fec1, "snr_db": snr, "sent": nf * N, "oracle_frames": len(o.frames),
                "oracle_valid": sum(1 for f in o.frames if f.header_valid and f.payload_valid),
                "oracle_equal_to_sent": sum(1 for f in o.frames if f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)),
                "gpu_frames": len(rx.frames)}
        try:
            case["worst_symbol_error"] = float(check_frames(rx.frames, o.frames)) if o.frames or rx.frames else 0.0
            case["gpu_equals_oracle"] = True
        except AssertionError as e:
            case["gpu_equals_oracle"] = False
            case["why"] = str(e)[:300]
        rx.close()
        out["cases"].append(case)
    print("VARIANT " + json.dumps(out))
    return 0 if all(c["gpu_equals_oracle"] for c in out["cases"]) else 1


if __name__ == "__main__":
    sys.exit(main())
