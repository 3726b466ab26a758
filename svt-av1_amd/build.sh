#!/bin/bash
# Builds the engine's shared library for gfx950 (cross-compiles without a GPU) and the two plain-C programs that link against it.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
make -s -C "$HERE" -j"${MAX_JOBS:-8}" all
