#!/bin/bash
# usage: tools/exp_build_run.sh "<extra -D flags>" : rebuild the library with experiment flags (timing-only)
set -e
cd "$(dirname "$0")/.."
touch nlml_hpe_amd/csrc/encoder_heads.hip
make -s -C nlml_hpe_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function $1" 2>&1 | grep -E "error" || true
