#!/bin/bash
# Builds an alternative liblec_hip.so with extra defines (measurement variants; loaded through LEC_LIB=<path>).
# Usage: tools/build_variant.sh <name> "<-D flags>"   ->  tools/probes/liblec_<name>.so
set -e
NAME=$1; DEFS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/lec_variant_$NAME; rm -rf $B; mkdir -p $B
cd $ROOT/lorenzcycletoolkit_amd/csrc
for f in lec_common lec_rowstats lec_rowsweep lec_rowblock lec_boxtile lec_boxplane lec_reduce lec_ingest lec_diag lec_check lec_hostio lec_inflate lec_format; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Werror=pass-failed $DEFS -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probes/liblec_$NAME.so $B/*.o
echo built tools/probes/liblec_$NAME.so
