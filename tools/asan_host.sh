#!/bin/bash
# The library's HOST code under AddressSanitizer (GPU ASan is not available on this pool): every
# csrc/*.hip compiled with -Xarch_host -fsanitize=address into a scratch copy of the package, then
# the file fuzzer and the CPU test files that call into the library run against that copy.
#   tools/asan_host.sh [fuzz cases]        (CPU only; about two minutes)
#   SANITIZERS=address,undefined tools/asan_host.sh      adds UBSan (reports go to stderr: grep "runtime error")
set -e
cases=${1:-4000}
sanitizers=${SANITIZERS:-address}
repo=$(cd "$(dirname "$0")/.." && pwd)
scratch=${TMPDIR:-/tmp}/emphases_asan
rm -rf $scratch && mkdir -p $scratch/build
cp -r $repo/emphases_amd $repo/tests $repo/oracle $repo/include $repo/tools $scratch/
cp $repo/BASELINE.json $repo/bench.py $scratch/ 2>/dev/null || true
for source in $repo/emphases_amd/csrc/*.hip; do
    name=$(basename $source .hip)
    /opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -fPIC -std=c++17 -Xarch_host -fsanitize=$sanitizers -Xarch_host -fno-sanitize=vptr,function \
        -Xarch_host -fno-omit-frame-pointer -I$repo/include -I$repo/emphases_amd/csrc \
        -c $source -o $scratch/build/$name.o &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$sanitizers $scratch/build/*.o -lpthread \
    -o $scratch/emphases_amd/libemphases_hip.so
runtime=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
cd $scratch
export UBSAN_OPTIONS=print_stacktrace=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$runtime
python tests/fuzz_files.py $cases 31337
python -m pytest tests/test_host.py tests/test_oracle.py -x -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_host.py::test_hand_scheduled_loads_are_not_touched_in_flight
