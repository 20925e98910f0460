#!/bin/bash
# Builds libTransform360.so a second time from a copy of the sources at another path and compares the bytes with the in-tree
# build (transform360_amd/csrc/Makefile: -ffile-prefix-map, relative include paths, a fixed -cuid per file, no build id).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)/some/other/place
mkdir -p $T/transform360_amd && cp -r $R/include $T/ && cp -r $R/transform360_amd/csrc $T/transform360_amd/ && rm -rf $T/transform360_amd/csrc/build
make -C $T/transform360_amd/csrc -j8 > /dev/null 2>&1
make -C $R/transform360_amd/csrc -j8 > /dev/null 2>&1
a=$(sha256sum < $R/transform360_amd/lib/libTransform360.so); b=$(sha256sum < $T/transform360_amd/lib/libTransform360.so)
echo "in tree: $a"; echo "copy:    $b"
[ "$a" = "$b" ] && echo "reproducible" || { echo "NOT reproducible"; exit 1; }
