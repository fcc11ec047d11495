#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh NAME "-DFOO -DBAR"
# -> transhuman_amd/_variants/libNAME.so  (use with TH_LIB_PATH=...)
set -e
cd "$(dirname "$0")/../transhuman_amd"
mkdir -p _variants/_obj_$1
for f in csrc/*.hip; do
  o=_variants/_obj_$1/$(basename ${f%.hip}).o
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../include -Icsrc $2 -c $f -o $o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 _variants/_obj_$1/*.o -o _variants/lib$1.so
rm -rf _variants/_obj_$1
echo built _variants/lib$1.so
