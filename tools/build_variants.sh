#!/bin/bash
# Build the product library and (ab/libstamps.so) a copy with in-kernel phase stamps; run here, before gpurun.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/ffmpeg-heaac_amd/csrc -j8 -s
rm -rf /tmp/csrc_stamps && mkdir -p /tmp/csrc_stamps $root/ab
cp $root/ffmpeg-heaac_amd/csrc/*.hip $root/ffmpeg-heaac_amd/csrc/*.h $root/ffmpeg-heaac_amd/csrc/*.c $root/ffmpeg-heaac_amd/csrc/Makefile /tmp/csrc_stamps/
make -C /tmp/csrc_stamps -j8 -s ROOT=$root EXTRA="-DHF_STAMPS -DPS_STAMPS" OUT=$root/ab/libstamps.so
