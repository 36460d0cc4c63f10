#!/bin/bash
# usage: tools/sanitize_host.sh [asan|tsan]   (CPU only, no GPU needed)
# Builds the host library (host/*.cpp: FASTQ / gzip / BAM readers, coercion, the multi-GPU driver) with
# AddressSanitizer + UBSan, or ThreadSanitizer, into /tmp/nimble_<kind>/lib beside a copy of the ordinary device library,
# and runs the CPU test suite against it (NIMBLE_LIB_DIR).  Sanitizers cannot run on the GPU side on this pool; the device
# library is exercised by the GPU parity suites instead.
KIND=${1:-asan}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/nimble_$KIND
case $KIND in
  asan) FLAGS="-fsanitize=address,undefined"; RT=$(gcc -print-file-name=libasan.so)
        export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 ;;
  tsan) FLAGS="-fsanitize=thread"; RT=$(gcc -print-file-name=libtsan.so)
        export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ;;
  *) echo "asan or tsan"; exit 2 ;;
esac
mkdir -p $OUT/lib $OUT/obj
cd "$ROOT/nimble-aligner_amd" || exit 1
make > /dev/null || exit 1
OBJS=""
for f in reference_library align fastq pgzip bam multi host_capi; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off $FLAGS -fno-omit-frame-pointer -c host/$f.cpp -o $OUT/obj/$f.o 2>&1 | grep -E "error:" && exit 1
  OBJS="$OBJS $OUT/obj/$f.o"
done
# the index builder (csrc/flat_index.cpp, plain C++ inside the device library) instrumented too; the HIP objects as built
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off $FLAGS -fno-sanitize=vptr -fno-omit-frame-pointer -pthread -c csrc/flat_index.cpp -o $OUT/obj/flat_index.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/lib/libnimble_hip.so build/kernels.o build/capi.o $OUT/obj/flat_index.o \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-z,undefs || exit 1
[ $KIND = asan ] && RT="$RT $(gcc -print-file-name=libubsan.so)"
g++ -shared $FLAGS -o $OUT/lib/libnimble_host.so $OBJS -L$OUT/lib -lnimble_hip -lz -pthread -Wl,-rpath,'$ORIGIN' || exit 1
cd "$ROOT"
# (libstdc++ beside the runtime: its __cxa_throw interceptor needs the real one resolvable when python loads the library)
NIMBLE_LIB_DIR=$OUT/lib LD_PRELOAD="$RT /usr/lib/x86_64-linux-gnu/libstdc++.so.6" \
  python -m pytest ${SAN_TESTS:-tests} -q -s -m "not gpu" > $OUT/out.txt 2>&1
echo "pytest rc=$?"
grep -a "runtime error\|ERROR: AddressSanitizer\|WARNING: ThreadSanitizer\|passed\|failed" $OUT/out.txt | sort | uniq -c
