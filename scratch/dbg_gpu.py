import sys, os, faulthandler
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import oracle as O
from __graft_entry__ import load_product
P = load_product()
N, M, cp, mod, fec1, plen, nf = [int(v) for v in sys.argv[1:8]]
iq, sent = O.synth_traffic(N, M, cp, 4, nf, payload_len=plen, mod=mod, fec1=fec1)
ora = O.MultiChannelRx(N, M, cp, 4); ora.execute(iq)
print("oracle frames", len(ora.frames), flush=True)
K = 2 * N
nb = len(iq) // K // 8 * 8
rx = P.multichannelrx(N, M, cp, 4, max_payload_len=max(plen, 64))
d_x = torch.from_numpy(iq[:nb * K]).cuda()
d_out = torch.zeros(nb * N, dtype=torch.complex64, device='cuda')
rx.channelize(d_x, nb, 0, d_out)
torch.cuda.synchronize(); print("channelize ok", flush=True)
ref = ora_ch = O.MultiChannelRx(N, M, cp, 4).channelize(iq[:nb * K])
got = P.tiles_to_channels(d_out, N).T
print("chan relerr", np.max(np.abs(got - ref)) / np.max(np.abs(ref)), flush=True)
rx.sync(d_out, 0, nb)
torch.cuda.synchronize(); print("sync ok", flush=True)
rx.Flush()
print(rx.frames, flush=True)
for fg, fo in zip(sorted(rx.frames, key=lambda f: (f.channel, f.end_sample)), sorted(ora.frames, key=lambda f: f.channel)):
    e = np.max(np.abs(fg.framesyms - fo.framesyms)) / np.max(np.abs(fo.framesyms)) if len(fg.framesyms) == len(fo.framesyms) and len(fo.framesyms) else -1
    print(fg.channel, fg.header_valid, fg.payload_valid, fg.payload == fo.payload, "relerr", e, fg.evm, fo.evm, fg.rssi, fo.rssi, fg.cfo, fo.cfo)
