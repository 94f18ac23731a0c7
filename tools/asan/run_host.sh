#!/bin/bash
# CPU only: the host layer (OBJ reader, .pts reader, staging, Triangle::Init, flattening, the host BVH builder, the texture decoders) built
# with AddressSanitizer + UBSan against stubs of the device layer, and fed malformed OBJ and .pts files (out-of-range / zero / huge
# indices, garbage tokens, damaged numbers, cut-short files).   bash tools/asan/run_host.sh [count]     Round 3: 2 000 files, no finding.
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; W=${TMPDIR:-/tmp}/ptk_asan_host; rm -rf $W; mkdir -p $W; cd $W
C=$ROOT/pbrpathtracer_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -I$ROOT/include -I$C $C/pathtracer.cpp $C/scene_io.cpp $C/bvh_build.cpp $C/image.cpp \
    $ROOT/tools/asan/device_stubs.cpp $ROOT/tools/asan/host_harness.cpp -lz -o host_asan || exit 1
python3 $ROOT/tools/asan/malformed_scenes.py bad ${1:-500} > /dev/null
ls bad/[0-9]*.obj bad/[0-9]*.pts | xargs -n 100 timeout 600 ./host_asan 2>&1 | grep -v "^triangles" | head -40
echo "done (anything above this line is a sanitizer finding)"
