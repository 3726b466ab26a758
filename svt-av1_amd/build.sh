#!/bin/bash
# Builds the engine's shared library for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/svtme/libsvtme.so"
SRC="$HERE/csrc/engine.cpp $HERE/csrc/controls.cpp $HERE/csrc/store_kernels.hip $HERE/csrc/hme_kernel.hip $HERE/csrc/hme_items.hip $HERE/csrc/me_kernel.hip $HERE/csrc/tail_kernel.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    ${SVTME_EXTRA_FLAGS} -o "$OUT" $SRC
echo "built $OUT"
