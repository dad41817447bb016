#!/bin/bash
# run the unchanged multichannel_txrx app in loopback, then decode the captured stream with the oracle and the GPU receiver
make -C liquid-usrp_amd/host -s
MCTX_LOOPBACK=1 MCTX_TEE_FILE=/tmp/tee.bin timeout 120 liquid-usrp_amd/lib/multichannel_txrx_ref -n 4 -M 64 -C 8 -T 4 -P 400 > gpurun_out/txrx.out 2> gpurun_out/txrx.err
echo rc=$?
grep -c "transmitting packet" gpurun_out/txrx.out
grep -c "header:pass" gpurun_out/txrx.out
grep -c "header:FAIL" gpurun_out/txrx.out
tail -7 gpurun_out/txrx.out
head -c 600 gpurun_out/txrx.err
python scratch/txrx_cmp.py 2>&1 | grep -v Traceback | tail -8
