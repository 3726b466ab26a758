#!/bin/bash
# Host-side sanitizer run of the C ABI (needs a GPU): builds the library with AddressSanitizer and
# UndefinedBehaviorSanitizer on the HOST code only (device code is compiled as usual), links tools/san_driver.c against
# it and runs it. Usage: tools/run_sanitized.sh [build|run|all]   (build works without a GPU)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/svt-av1_amd/svtme/san"
C="$ROOT/svt-av1_amd/csrc"
RT="$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")"
what="${1:-all}"
if [ "$what" != run ]; then
    mkdir -p "$OUT"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
        -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
        -shared-libsan -o "$OUT/libsvtme.so" \
        "$C/engine.cpp" "$C/group.cpp" "$C/controls.cpp" "$C/store_kernels.hip" "$C/hme_kernel.hip" "$C/hme_items.hip" "$C/hme_regions.hip" "$C/me_kernel.hip" "$C/tail_kernel.hip" "$C/gm_kernels.hip" "$C/tpl_kernels.hip"
    /opt/rocm/lib/llvm/bin/clang -std=gnu99 -O1 -g -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=undefined \
        -shared-libsan -I "$ROOT/include" "$ROOT/tools/san_driver.c" -o "$OUT/san_driver" \
        -L "$OUT" -lsvtme -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$RT"
    echo "built $OUT/san_driver"
fi
if [ "$what" != build ]; then
    # the HIP runtime maps memory inside ASan's shadow gap; its own allocations at exit are not the library's leaks
    ASAN_OPTIONS="protect_shadow_gap=0:detect_leaks=${SVTME_SAN_LEAKS:-0}:abort_on_error=1" \
    UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" "$OUT/san_driver"
fi
