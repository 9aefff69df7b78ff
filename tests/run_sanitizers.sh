#!/bin/bash
# AddressSanitizer + UBSan pass over the CPU-side code (never on the GPU box: sanitizers and GPU runs do not mix on this pool).
# Builds the sanitized variants of the oracle, the host simulation (host SAH builder, two-level build, scene flattening, the device
# functions compiled for the host) and the host layer (VSGF + Hydra XML readers), then runs the whole CPU suite against them.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle ASAN=1
make -s -C tests/host_sim ASAN=1
make -s -C ada-ray-tracer_amd libart_host_asan.so
export ART_ASAN=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
exec python -m pytest tests -q -m "not gpu" "$@"
