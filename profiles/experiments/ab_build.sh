#!/bin/bash
# Build one named variant of libsr_engine.so for a same-box A/B (profiles/experiments/ab.py).
#   profiles/experiments/ab_build.sh NAME [GIT_REF] [EXTRA_CXXFLAGS...]
# NAME.so lands in ab_libs/ (git-ignored as *.so, but it travels to the GPU box with gpurun's snapshot).
#   GIT_REF   commit / tag / branch whose csrc + include are built ("." or empty = the working tree)
#   EXTRA     e.g. -DSR_DTW_STATS ; appended to the Makefile's CXXFLAGS
# The A/B itself runs on ONE box, alternating the builds (boxes differ by up to 3 % in shader clock, far more than most
# of the steps being compared):
#   gpurun -- 'python profiles/experiments/ab.py ab_libs/base.so ab_libs/new.so 3 [--gain 2.4] [--workload ext]'
set -euo pipefail
name=$1; ref=${2:-.}; shift; shift || true
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$root/ab_libs"
work=$(mktemp -d /tmp/ab_build.XXXXXX)
trap 'rm -rf "$work"' EXIT
if [ "$ref" = "." ]; then
    mkdir -p "$work/stm32_speech_recognition_amd" "$work/include"
    cp -r "$root/stm32_speech_recognition_amd/csrc" "$work/stm32_speech_recognition_amd/csrc"
    cp "$root/include/sr_engine.h" "$work/include/"
    rm -rf "$work/stm32_speech_recognition_amd/csrc/build" "$work/stm32_speech_recognition_amd/csrc/build_testing"
else
    git -C "$root" archive "$ref" stm32_speech_recognition_amd/csrc include | tar -x -C "$work"
fi
flags="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $*"
make -C "$work/stm32_speech_recognition_amd/csrc" -j8 ../libsr_engine.so CXXFLAGS="$flags" >/dev/null
cp "$work/stm32_speech_recognition_amd/libsr_engine.so" "$root/ab_libs/$name.so"
echo "ab_libs/$name.so  <-  ${ref}  ${*:-}"
