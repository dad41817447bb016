"""Handles belong to the device they were created on (csrc/devscope.hpp): the one-process-many-GPU host.
With one GPU in the box the device switch itself cannot happen; what can be checked there is that the handle knows its device and that
every kind of handle works when created and used with that device current.  With two or more, the caller's current device is moved away
between creation and use, and handles are opened on the second device after the first -- where the per-process statics of rounds 1-5
skipped the dynamic-LDS setting and a 127 KB-LDS launch failed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_receiver(product, oracle, torch, device, switch_to=None):
    N, M, cp = 512, 64, 8                                   # K = 1024: the 127 KB-LDS channelizer instantiation
    with torch.cuda.device(device):
        tx = product.multichanneltx(N, M, cp, 4)
        iq, sent = tx.generate(1, 64, seed=3)
        rx = product.multichannelrx(N, M, cp, 4, max_payload_len=64)
        assert product.lib().mcrx_hip_device(rx._h) == device
    if switch_to is not None:
        torch.cuda.set_device(switch_to)                    # the caller's current device is now another one
    n = int(iq.numel()) // (32 * N) * (32 * N)
    rx.Execute(iq[:n]); rx.Flush()
    assert len(rx.frames) == N
    for f in rx.frames:
        assert f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    if switch_to is not None:
        assert torch.cuda.current_device() == switch_to    # ... and still is: every entry point put it back
    rx.close(); tx.close()


def test_handle_knows_its_device_and_runs_there(product, oracle):
    import torch
    assert torch.cuda.is_available()
    _run_receiver(product, oracle, torch, 0)


def test_handles_on_a_second_device_and_a_caller_that_moves(product, oracle):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU in this box: the device switch needs two (the bookkeeping itself: tests/test_boundary.py)")
    _run_receiver(product, oracle, torch, 0)
    _run_receiver(product, oracle, torch, 1)                # second device of the same process: its own LDS limits
    _run_receiver(product, oracle, torch, 0, switch_to=1)   # created on 0, used while 1 is current
    torch.cuda.set_device(0)
