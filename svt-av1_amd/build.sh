#!/bin/bash
# Builds the engine's shared library for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/svtme/libsvtme.so"
SRC="$HERE/csrc/engine.cpp $HERE/csrc/group.cpp $HERE/csrc/controls.cpp $HERE/csrc/store_kernels.hip $HERE/csrc/hme_kernel.hip $HERE/csrc/hme_items.hip $HERE/csrc/hme_regions.hip $HERE/csrc/me_kernel.hip $HERE/csrc/tail_kernel.hip $HERE/csrc/gm_kernels.hip $HERE/csrc/tpl_kernels.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    ${SVTME_EXTRA_FLAGS} -o "$OUT" $SRC
echo "built $OUT"
# the plain-C example of the reference-side binding (links only against the C ABI)
gcc -std=c99 -O2 -Wall -Wextra -I "$HERE/../include" "$HERE/examples/me_shim_demo.c" -o "$HERE/svtme/me_shim_demo" \
    -L "$HERE/svtme" -lsvtme -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo "built $HERE/svtme/me_shim_demo"
# the plain-C life-cycle / error-path driver (also the program tools/run_sanitized.sh runs against a host-sanitized build)
gcc -std=gnu99 -O2 -Wall -Wextra -I "$HERE/../include" "$HERE/../tools/san_driver.c" -o "$HERE/svtme/abi_stress" \
    -L "$HERE/svtme" -lsvtme -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo "built $HERE/svtme/abi_stress"
