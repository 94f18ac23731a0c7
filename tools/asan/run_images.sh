#!/bin/bash
# CPU only: the texture decoders (csrc/image.cpp) built with AddressSanitizer + UBSan and fed corrupted copies of every image fixture
# (flipped bytes, cut-short files, overwritten runs, damaged header fields).   bash tools/asan/run_images.sh [count]
# Allocations over 3 GB are sanitizer errors (ASAN_OPTIONS=max_allocation_size_mb): a decoder that believes a damaged size field and
# allocates for it is a finding, and so is a timeout.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; W=${TMPDIR:-/tmp}/ptk_asan_images; rm -rf $W; mkdir -p $W; cd $W
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include -I$ROOT/pbrpathtracer_amd/csrc $ROOT/pbrpathtracer_amd/csrc/image.cpp $ROOT/tools/asan/image_harness.cpp -lz -o img_asan
python3 $ROOT/tools/asan/corrupt_images.py corrupt 0 ${1:-1000} > /dev/null
set +e
export ASAN_OPTIONS=max_allocation_size_mb=3000:allocator_may_return_null=0
ls corrupt > list.txt; split -l 40 list.txt batch_; flagged=0
for b in batch_*; do
  if ! timeout 40 ./img_asan $(sed 's#^#corrupt/#' $b | tr '\n' ' ') > out.txt 2>&1; then
    while read f; do
      timeout 8 ./img_asan corrupt/$f > out1.txt 2>&1; r=$?
      # a timeout or an abort (rc 124 / 134: an allocation the header talked the decoder into) is a FINDING, not "slow" 
      if [ $r -ne 0 ]; then echo "rc=$r $f: $(grep -m2 -E 'ERROR|runtime error|bad_alloc|terminate' out1.txt | cut -c1-200 | tr '\n' ' ')"; flagged=$((flagged+1)); fi
    done < $b
  fi
done
echo "sanitizer findings: $flagged"
