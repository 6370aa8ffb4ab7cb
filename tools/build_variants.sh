#!/bin/bash
# Build the product library and variants of it under ab/ (run here, before gpurun):
#   tools/build_variants.sh                      product + ab/libstamps.so (in-kernel phase stamps)
#   tools/build_variants.sh name "-DFOO -DBAR"   additionally ab/libname.so with those extra flags
#   VARIANT_SED='s/a/b/' tools/build_variants.sh name ""   ... with a sed script applied to the copied sources
#   VARIANT_EDIT=path/to/edit.py tools/build_variants.sh name ""   ... with `python3 edit.py <copied dir> <name>` run on them
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/ffmpeg-heaac_amd/csrc -j8 -s
mkdir -p $root/ab
build() {   # name, flags, sed script
    rm -rf /tmp/csrc_$1 && mkdir -p /tmp/csrc_$1
    cp $root/ffmpeg-heaac_amd/csrc/*.hip $root/ffmpeg-heaac_amd/csrc/*.h $root/ffmpeg-heaac_amd/csrc/*.c $root/ffmpeg-heaac_amd/csrc/Makefile /tmp/csrc_$1/
    if [ -n "$3" ]; then sed -i -E "$3" /tmp/csrc_$1/*.hip /tmp/csrc_$1/*.h; fi
    if [ -n "$VARIANT_EDIT" ]; then python3 "$VARIANT_EDIT" /tmp/csrc_$1 "$1"; fi
    make -C /tmp/csrc_$1 -j8 -s ROOT=$root EXTRA="$2" OUT=$root/ab/lib$1.so
}
if [ -n "$1" ]; then build "$1" "$2" "$VARIANT_SED"; else build stamps "-DHEAAC_STAMPS" ""; fi
