#!/bin/bash
# tools/lab/build_variant.sh NAME "-DFOO=1 -DBAR=2" [files...]: a second build of libhipblosc.so with extra macros on the named translation
# units (default: the two LZ4 kernels' files), linked with the standard objects of the others -> go-blosc_amd/lib/libhipblosc_NAME.so.
# Lab tooling for tools/lab/ab.py (kernel variants timed side by side in ONE gpurun call); not part of the product build.
set -e
NAME=$1; EXTRA=$2; shift 2 || true
FILES=${@:-"hb_lz4_enc.hip hb_lz4_dec.hip"}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/go-blosc_amd/csrc
OD=$ROOT/scratch/var/$NAME
mkdir -p "$OD"
make -C "$CS" -j4 -s >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$ROOT/include $EXTRA"
OBJS=""
pids=""
for f in $FILES; do
  hipcc $FLAGS -c "$CS/$f" -o "$OD/${f%.hip}.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for o in "$CS"/*.o; do
  b=$(basename "$o")
  if [ -f "$OD/$b" ]; then OBJS="$OBJS $OD/$b"; else OBJS="$OBJS $o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/go-blosc_amd/lib/libhipblosc_$NAME.so" $OBJS -ldl -lpthread
echo "built libhipblosc_$NAME.so ($EXTRA)"
