#!/bin/bash
# run the unchanged fullduplex app in loopback until a frame goes missing; keep the recording of that run
make -C liquid-usrp_amd/host -s
for i in $(seq 1 ${1:-40}); do
  MCTX_LOOPBACK=1 MCTX_TEE_FILE=/tmp/fdx.bin liquid-usrp_amd/lib/fullduplex_txrx_ref -N 200 -P 500 -m qam16 -c h128 -k none > /tmp/fdx.out 2>/tmp/fdx.err
  n=$(grep -c "rx packet id" /tmp/fdx.out)
  echo "run $i: $n frames"
  if [ "$n" != "200" ]; then
    cp /tmp/fdx.out gpurun_out/fdx.out
    python scratch/fdx_cmp.py
    break
  fi
done
