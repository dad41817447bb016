#!/bin/bash
# make_ref_golden.sh -- pin-if-present: if a real liquid-dsp is installed, build the reference's OWN lib/multichanneltx.cc and
# lib/multichannelrx.cc (unchanged, from /root/reference) against it with tests/golden/ref_harness.cc as the driver, and emit fixtures
# (wideband IQ, the frames its callbacks received, their equalised symbols) for the shapes of BASELINE.json configs[0..2] into
# tests/golden/ref_liquid/.  tests/test_liquid_interop.py compares the CPU oracle and the GPU path with them.
#   exit 0: fixtures written      exit 3: no liquid-dsp / no reference tree here (nothing written)
# The binary goes to oracle/_ref/ (git-ignored); the fixtures are data and are meant to be committed.
# Where liquid is looked for: $LIQUID_PREFIX (include/ + lib/ under it), pkg-config liquid, then the compiler's default paths.
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${MCRX_REFERENCE_TREE:-/root/reference}"
OUT="$HERE/ref_liquid"
BIN="$ROOT/oracle/_ref"
[ -f "$REF/lib/multichannelrx.cc" ] || { echo "make_ref_golden: no reference tree at $REF" >&2; exit 3; }
CFLAGS=""; LIBS="-lliquid"
if [ -n "${LIQUID_PREFIX:-}" ]; then CFLAGS="-I$LIQUID_PREFIX/include"; LIBS="-L$LIQUID_PREFIX/lib -Wl,-rpath,$LIQUID_PREFIX/lib -lliquid"
elif pkg-config --exists liquid 2>/dev/null; then CFLAGS="$(pkg-config --cflags liquid)"; LIBS="$(pkg-config --libs liquid)"; fi
printf '#include <liquid/liquid.h>\nint main(){return liquid_libversion()[0]==0;}\n' > /tmp/mcrx_liquid_probe.c
if ! gcc $CFLAGS /tmp/mcrx_liquid_probe.c -o /tmp/mcrx_liquid_probe $LIBS -lm 2>/dev/null; then
    echo "make_ref_golden: no liquid-dsp found (LIQUID_PREFIX, pkg-config liquid, default paths): nothing to pin against" >&2; exit 3
fi
mkdir -p "$OUT" "$BIN"
g++ -O2 -std=c++11 -I"$REF/include" $CFLAGS "$HERE/ref_harness.cc" "$REF/lib/multichanneltx.cc" "$REF/lib/multichannelrx.cc" \
    -o "$BIN/ref_harness" $LIBS -lm -lpthread || { echo "make_ref_golden: the reference's classes did not build against this liquid-dsp" >&2; exit 1; }
set -e
#                 prefix                    N   M  cp taper mod   fec0 fec1  plen frames seed
"$BIN/ref_harness" "$OUT/c1_1ch_m64_qpsk"    1  64  8  4    qpsk  none h128  1200  3     1
"$BIN/ref_harness" "$OUT/c2_8ch_m64_qpsk"    8  64  8  4    qpsk  none h128  1200  2     2
"$BIN/ref_harness" "$OUT/c3_4ch_m256_qam16"  4 256 32  4    qam16 none g2412 1200  2     3
"$BIN/ref_harness" "$OUT/x_2ch_m48_bpsk"     2  48  6  4    bpsk  none none   100  2     4
"$BIN/ref_harness" "$OUT/x_2ch_m64_v27"      2  64  8  4    qpsk  none v27    300  2     5
echo "make_ref_golden: fixtures in $OUT"
