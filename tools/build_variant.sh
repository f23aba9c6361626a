#!/bin/bash
# A second build of the library with extra compiler flags, for A/B timing through PANO_LIB:
#   tools/build_variant.sh NAME "-DMB_SOMETHING=1 ..."   ->  build/variants/NAME/libpano360_hip.so
# (build/ is git-ignored but travels to the GPU box with the snapshot)
set -eu
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=${2:-}
DST=build/variants/$NAME
mkdir -p "$DST/csrc"
cp pano360_amd/csrc/*.hip pano360_amd/csrc/*.h pano360_amd/csrc/*.inc pano360_amd/csrc/Makefile "$DST/csrc/"
# the sources include ../../include/pano360.h relative to csrc/
mkdir -p build/variants/include && cp include/pano360.h build/variants/include/
make -s -C "$DST/csrc" -j8 EXTRA="$EXTRA" OUT=../libpano360_hip.so
ls -la "$DST/libpano360_hip.so"
