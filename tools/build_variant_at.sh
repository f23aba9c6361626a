#!/bin/bash
# The library as of a commit, for A/B timing through PANO_LIB against the working tree's:
#   tools/build_variant_at.sh REV NAME ["-DFLAGS"]   ->  build/variants/NAME/libpano360_hip.so
set -eu
cd "$(dirname "$0")/.."
REV=$1; NAME=$2; EXTRA=${3:-}
DST=build/variants/$NAME
rm -rf "$DST"; mkdir -p "$DST/csrc" build/variants/include
for f in $(git ls-tree --name-only "$REV" pano360_amd/csrc/); do
  git show "$REV:$f" > "$DST/csrc/$(basename "$f")"
done
git show "$REV:include/pano360.h" > build/variants/include/pano360.h
make -s -C "$DST/csrc" -j8 EXTRA="$EXTRA" OUT=../libpano360_hip.so
cp include/pano360.h build/variants/include/pano360.h
ls -la "$DST/libpano360_hip.so"
