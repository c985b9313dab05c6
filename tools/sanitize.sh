#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host code (SURVEY §5 "Race detection / sanitizers").
#   tools/sanitize.sh build   builds tools/variants/libpinot_gpu_san.so (host objects instrumented, device code as always) and
#                             oracle/_build/liboracle_san.so, both with clang so that ONE runtime serves the process
#   tools/sanitize.sh cpu     the whole `not gpu` suite with the instrumented oracle (and the JNI shim's parser tests rebuilt with -fsanitize)
#   tools/sanitize.sh gpu     (GPU box) the malformed-buffer fuzz + the parity suites against the instrumented library
# Output: gpurun_out/r06_sanitize_<mode>.log; a clean run has no "ERROR: AddressSanitizer" / "runtime error:" line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
case "${1:-cpu}" in
  build)
    bash tools/build_variants.sh san "-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g" | tail -2
    make -s -C oracle san ;;
  cpu)
    make -s -C oracle san
    CC=/opt/rocm/lib/llvm/bin/clang
    for t in jni_sequence_test; do   # the NativeQuery record parser (truncation sweep, bad magic) under the sanitizers
      $CC -std=c99 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iintegration/jni integration/jni/$t.c integration/jni/pinot_gpu_shim.c \
        -Lpinot_amd/csrc -lpinot_gpu -Wl,-rpath,$R/pinot_amd/csrc -o /tmp/${t}_san && /tmp/${t}_san 2>&1 | tail -3
    done > gpurun_out/r06_sanitize_cpu.log 2>&1
    LD_PRELOAD=$RT PO_ORACLE_LIB=$R/oracle/_build/liboracle_san.so timeout 3000 python -m pytest tests -q -m "not gpu" -p no:cacheprovider >> gpurun_out/r06_sanitize_cpu.log 2>&1
    grep -c "ERROR: AddressSanitizer\|runtime error:" gpurun_out/r06_sanitize_cpu.log; tail -3 gpurun_out/r06_sanitize_cpu.log ;;
  gpu)
    export LD_PRELOAD=$RT PG_GPU_LIB=$R/tools/variants/libpinot_gpu_san.so PO_ORACLE_LIB=$R/oracle/_build/liboracle_san.so
    timeout 1500 python tests/malformed_worker.py > gpurun_out/r06_sanitize_gpu.log 2>&1
    timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition_pipeline.py tests/test_gpu_group_trim.py tests/test_segment_dir.py tests/test_gpu_startree.py \
      tests/test_gpu_mv.py tests/test_gpu_datatable.py tests/test_null_handling_filters.py tests/test_gpu_multi.py tests/test_gpu_dict_headline.py tests/test_gpu_mv_group.py tests/test_distinctcount_raw.py tests/test_gpu_filter_stats_device.py tests/test_null_handling_trim.py tests/test_null_handling_aggregations.py tests/test_raw_string_predicates.py tests/test_gpu_fake_rccl.py -x -q -m gpu -p no:cacheprovider >> gpurun_out/r06_sanitize_gpu.log 2>&1
    grep -c "ERROR: AddressSanitizer\|runtime error:" gpurun_out/r06_sanitize_gpu.log; tail -4 gpurun_out/r06_sanitize_gpu.log ;;
esac
